"""Training-step harness around the model: per-step Bernoulli node mask,
class-weighted cross entropy, SGD with momentum over one flat bucket, and tree-sharded data
parallelism with a single RCCL all-reduce per step.

Reference: job_runner.py:1886-1920 (SPGNN) / 1393-1416 (GCN/GAT/GIN).  Differences, all host-side:
the mask and the loss are evaluated on device without ``nonzero`` syncs, and with W > 1 ranks the
class-weighted mean is normalised by the GLOBAL weight sum (each rank back-propagates its local
weighted SUM; the weight sum rides in the same all-reduce bucket), so W ranks x B trees take exactly
the step one rank would take on W*B trees.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import ops

__all__ = ["sampling_probabilities", "mask_from_draws", "weighted_nll_sums", "masked_weighted_ce", "FlatBucket",
           "TrainStep", "balanced_tree_partition"]


def balanced_tree_partition(node_counts: Sequence[int], world: int) -> List[List[int]]:
    """Shard the trees of one global batch over ``world`` ranks balancing the NODE count per rank (SURVEY.md §8e:
    "balance by sum n_i, not tree count" - the work of every kernel is proportional to a rank's nodes and edges, and
    E = 3N - 2B).  Longest-processing-time greedy: trees in descending size, each to the lightest rank so far (ties: the
    lower rank); within a rank the trees keep their order of the global batch.  For airway batches (hundreds of trees of
    100-300 nodes per rank) the per-rank sums differ by well under 1 %.  Deterministic: every rank computes the same
    partition from the same list."""
    if world <= 0:
        raise ValueError("world must be positive")
    order = sorted(range(len(node_counts)), key=lambda i: (-int(node_counts[i]), i))
    load = [0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        parts[r].append(i)
        load[r] += int(node_counts[i])
    return [sorted(p) for p in parts]


def sampling_probabilities(labels: torch.Tensor, sampling_rate: float) -> torch.Tensor:
    """1.0 for labelled nodes (y != 0), SAMPLING_RATE for the rest (job_runner.py:1887-1888)."""
    p = torch.full(labels.shape, float(sampling_rate), dtype=torch.float32, device=labels.device)
    p[labels != 0] = 1.0
    return p


def mask_from_draws(draws: torch.Tensor, sampling_p: torch.Tensor) -> torch.Tensor:
    """``mask[i] = rn[i] < sampling_t[i]`` (job_runner.py:1896)."""
    return draws < sampling_p


def weighted_nll_sums(logits: torch.Tensor, labels: torch.Tensor, mask: torch.Tensor, class_weight: torch.Tensor):
    """(sum_i m_i w[y_i] * nll_i, sum_i m_i w[y_i]): numerator and denominator of
    ``F.cross_entropy(logits[mask], y[mask], weight=w)`` (job_runner.py:1900), without boolean indexing."""
    logp = F.log_softmax(logits.float(), dim=1)
    nll = -logp.gather(1, labels.view(-1, 1)).squeeze(1)
    w = class_weight[labels] * mask.to(logp.dtype)
    return (w * nll).sum(), w.sum()


def masked_weighted_ce(logits, labels, mask, class_weight) -> torch.Tensor:
    num, den = weighted_nll_sums(logits, labels, mask, class_weight)
    return num / den


class FlatBucket:
    """All trainable parameters, their gradients and momentum buffers as three flat fp32 tensors.
    ``p.data`` / ``p.grad`` become views, so autograd accumulates straight into the bucket and one
    all-reduce + one fused SGD launch cover the whole model.  Two extra slots behind the gradients carry the step's
    class-weight sum and loss numerator, so the exchange between ranks is exactly ONE collective."""

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        # every parameter starts on a 16-byte boundary of the bucket (a one-element parameter such as GINConv's eps would
        # otherwise leave every later weight and bias misaligned for the kernels' 16-byte loads); the padding floats stay zero
        # in all three buffers (zero gradient -> zero momentum -> no update)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off                     # extent of the parameter region (padding included)
        total = self.numel + 4
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_mom = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            self.flat_param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + n].view_as(p.data)
            p.grad = self.flat_grad[off:off + n].view_as(p.data)
        self.loss_slot = self.flat_grad[self.numel:self.numel + 1]          # [loss numerator, class-weight sum]: the order
        self.wsum_slot = self.flat_grad[self.numel + 1:self.numel + 2]      # ops.masked_ce_sums writes them in
        self.sums_slot = self.flat_grad[self.numel:self.numel + 2]
        self.steps = 0

        self._views = [p.grad for p in self.params]
        self._zeros = {}

    def zero_grad(self):
        self.flat_grad.zero_()
        for p in self.params:            # keep the views attached (a foreign .grad would bypass the bucket)
            if p.grad is None or p.grad.data_ptr() < self.flat_grad.data_ptr():
                raise RuntimeError("parameter .grad was detached from the flat bucket")

    def detach_grads(self):
        """Before backward: with ``.grad`` unset autograd hands each parameter its gradient tensor as is - no zero
        fill of the bucket and no accumulate launch per parameter (44 adds of a few microseconds each per step)."""
        for p in self.params:
            p.grad = None

    def gather_grads(self):
        """After backward: one batched copy of all gradients into the flat bucket; ``.grad`` becomes the bucket view again."""
        parts = []

        def zeros(k):
            z = self._zeros.get(k)
            if z is None:
                z = self._zeros[k] = torch.zeros(k, dtype=torch.float32, device=self.flat_grad.device)
            return z
        for p in self.params:
            g = p.grad
            n = p.numel()
            parts.append(zeros(n) if g is None else g.reshape(-1))   # None: not reached by this loss (e.g. unused auxiliary heads)
            if n % 4:
                parts.append(zeros(4 - n % 4))                       # the padding up to the next parameter
        torch.cat(parts, out=self.flat_grad[:self.numel])
        for p, v in zip(self.params, self._views):
            p.grad = v


ONE_GRAPH_PER_STEP = True    # one process: the captured step is ONE HIP graph (two, split at the all-reduce, when ranks exchange)


class TrainStep:
    """One optimizer step on a static batched graph: mask -> forward -> loss -> backward ->
    [all-reduce] -> SGD(momentum)."""

    def __init__(self, model: torch.nn.Module, class_weights: Sequence[float], sampling_rate: float, lr: float,
                 momentum: float = 0.9, weight_decay: float = 0.0, process_group=None, seed: int = 0,
                 range_policy: str = "monitor", loss_rows_only: bool = False, always_exchange: bool = False):
        """``loss_rows_only``: run what follows the last aggregation - output-layer projection, head mean, classifier and their
        backward products - only on the rows the step's mask keeps (``F.cross_entropy(pre[mask], ...)``, reference
        job_runner.py:1896-1900: no other row reaches the loss or a gradient).  Same loss and gradients up to fp32 summation
        order; the model's forward then returns one row per kept node instead of one per node, so it is a switch of the training
        step, not of the model.  Used by the heads that fuse output layer and classifier (ops._GATAggFirstFn: the SPGNN nets); the
        others run as before.  ``"backward"``: the forward pass stays dense (``model(g)`` inside the step returns every row, as
        the reference's does) and only the BACKWARD products of that part run on the kept rows - the rows they skip are exactly
        zero in the dense step (their logit gradients are).  See :meth:`_loss_rows_cap`.
        ``always_exchange``: issue the step's all-reduce (and split the captured step in two graphs around it) even when the
        process group has ONE rank - the multi-rank step with nothing to add, which lets a one-GPU box exercise the RCCL path
        (bench.py, SPGNN_BENCH_FORCE_LAUNCH).
        ``range_policy``: what :meth:`run_batch` does with the GEMM range monitor: "monitor" only
        counts (``range_violations()``); "auto" switches the split GEMMs to their wide-range form (``ops.GEMM_WIDE``) for all
        later batches once an operand left the narrow envelope, dropping the captures recorded in the narrow form."""
        if range_policy not in ("monitor", "auto"):
            raise ValueError("range_policy must be 'monitor' or 'auto'")
        self.range_policy, self._violations_seen = range_policy, None      # None: the counter's value at the first run_batch (it is per device, not per step)
        self.model = model
        self.bucket = FlatBucket(list(model.parameters()))
        dev = self.bucket.flat_param.device
        self.class_weight = torch.tensor(list(class_weights), dtype=torch.float32, device=dev)
        self.sampling_rate, self.lr, self.momentum, self.weight_decay = sampling_rate, lr, momentum, weight_decay
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.exchange = self.world > 1 or bool(always_exchange and dist.is_available() and dist.is_initialized())
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self._captures, self._arenas = {}, {}       # id(graph) -> its HIP graph(s); size class -> arena.BatchArena
        self.max_arenas = 8                         # size classes kept (buffers + captured graphs each); beyond: least recently used out
        self._graph = self._graph_back = self._captured_graph = None
        self._verify_deferred = True                # the first step checks that no deferred gradient was read before it was filled
        self._lr_dev = None
        self._one = self._loss_out = None
        # seed of the mask stream a captured step draws inside its loss kernel (spgnn_masked_ce_step); eager steps use self.gen
        self._mask_seed = (int(seed) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & ((1 << 62) - 1)
        if loss_rows_only not in (False, True, "backward"):
            raise ValueError('loss_rows_only must be False, True or "backward"')
        self.loss_rows_only = loss_rows_only
        self._rows_cnt = None                       # (2,) int32 on the device: [rows kept by the last step, capacity-overflow flag]
        self._skipped = None                        # (1,) int32 on the device: steps the guarded SGD kernel did not apply (non-finite loss)
        self._rows_headroom, self._rows_gen = 1.0, 0    # capacity factor / generation of the per-graph capacity caches

    def _loss_rows_cap(self, g, p: torch.Tensor) -> int:
        """Capacity of the step's row list on ``g``: the kept count is (labelled nodes) + Binomial(others, rate) - mean plus
        eight standard deviations (about 1e-15 per step), 15 % on top when other batches will be loaded into the same buffers
        (a batch arena), rounded up to 256-row tiles.  0: the list would not be shorter than 0.8 N - the step runs dense.
        One host read per graph, cached on it.  A draw beyond the capacity cannot be repaired inside a captured step: the
        list kernel raises a flag, the loss of that step is NaN and the guarded optimizer kernel does not apply it
        (spgnn_sgd_momentum_step_guarded: parameters and momentum stay); :meth:`check_loss_rows` (called by run_batch /
        run_batches once per loader batch) then enlarges the capacities and drops the captures made with the old ones."""
        store = g.__dict__.setdefault("_loss_rows_cap", {})
        key = (self.sampling_rate, self._rows_gen)
        cap = store.get(key)
        if cap is None:
            pc = p.clamp(min=0.0).double()
            mu, var = float(pc.sum()), float((pc * (1.0 - pc)).sum())
            mu *= self._rows_headroom * (1.15 if getattr(g, "_stable_storage", False) else 1.0)
            want = mu + 8.0 * var ** 0.5 + 32.0
            cap = int(-(-want // 256) * 256)
            n = p.shape[0]
            cap = 0 if cap > 0.8 * n else cap
            store[key] = cap
        return cap

    def _arena_rows_check(self, g, p: torch.Tensor) -> None:
        """Called when a new batch was loaded into arena graph ``g`` (its refresh hook): the row-list capacity of the step
        captured on ``g`` came from an EARLIER batch of the size class, and a batch with more labelled nodes (they are always
        kept) may not fit it - every inner step of that loader batch would overflow and be skipped.  One host read
        per loader batch: mean + 8 sigma of THIS batch's kept count; beyond the capacity the capacity grows (15 % on top) and
        the capture made with the old one is dropped, so the batch's first step records a new one."""
        if not self.loss_rows_only:
            return
        store = g.__dict__.get("_loss_rows_cap")
        key = (self.sampling_rate, self._rows_gen)
        cap = store.get(key) if store else None
        if not cap:                                  # not computed yet, or 0: the step on this graph runs dense
            return
        pc = p.clamp(min=0.0).double()
        mu, var = torch.stack([pc.sum(), (pc * (1.0 - pc)).sum()]).tolist()
        want = mu + 8.0 * var ** 0.5 + 32.0
        if want <= cap:
            return
        new = int(-(-(want * 1.15) // 256) * 256)
        store[key] = 0 if new > 0.8 * p.shape[0] else new
        self._captures.pop(id(g), None)
        if self._captured_graph is g:
            self._graph = self._graph_back = self._captured_graph = None

    def check_loss_rows(self) -> int:
        """One host read: did a step's mask keep more rows than its list could hold since the last check?  Such a step had a
        NaN loss and was not applied (the guarded optimizer kernel).  Then: every capacity grows by half (recomputed per graph on
        its next step), the captures recorded with the old capacities are dropped, a warning names the lost steps.
        -> the number of steps skipped so far."""
        if self._rows_cnt is None:
            return 0
        import warnings
        if self._skipped is not None:                # ONE blocking read: [steps skipped, overflow flag]
            skipped, flag = torch.cat([self._skipped, self._rows_cnt[1:2]]).tolist()
        else:
            skipped, flag = 0, int(self._rows_cnt[1].item())
        if flag == 0:
            if skipped > getattr(self, "_skipped_seen", 0) and self.world == 1:      # (several ranks: another rank's list may have overflowed)
                # the guarded optimizer kernel skips EVERY step with a non-finite loss; without an overflow that is divergence,
                # a zero weight sum or a bad label - the reference would propagate the NaN, so say it
                warnings.warn(f"loss_rows_only: {skipped - getattr(self, '_skipped_seen', 0)} step(s) had a non-finite loss WITHOUT a "
                              "row-list overflow and were not applied: the training has diverged (or a batch has no kept node).",
                              RuntimeWarning)
            self._skipped_seen = skipped
            return skipped
        self._skipped_seen = skipped
        self._rows_cnt[1].zero_()
        self._rows_headroom *= 1.5
        self._rows_gen += 1
        # only captures whose graph carries a row list were recorded with a capacity; the others stay
        for k in [k for k, rec in self._captures.items() if any(rec["graph"].__dict__.get("_loss_rows_cap", {}).values())]:
            if self._captured_graph is self._captures[k]["graph"]:
                self._graph = self._graph_back = self._captured_graph = None
            del self._captures[k]
        warnings.warn(f"loss_rows_only: a step's mask kept more rows than its row list could hold; {skipped} step(s) so far had a NaN "
                      f"loss and were not applied.  The lists now get {self._rows_headroom:.2f} x the headroom; captured steps are "
                      "recorded again.", RuntimeWarning)
        return skipped

    def _sampling(self, g):
        """Per-node sampling probabilities of ``g``'s labels, kept ON THE GRAPH (a captured step addresses the tensor; one
        step object may hold captures on several graphs).  Pad nodes of a batch arena (spgnn_amd/arena.py) get -1: the mask
        ``rn < p`` of job_runner.py:1896 never keeps them.  On an arena the values are recomputed into the same storage
        whenever a batch is loaded (a refresh hook on the graph)."""
        y = g.ndata["y"]
        store = g.__dict__.setdefault("_sampling_p", {})
        hit = store.get(self.sampling_rate)
        if hit is None or hit[0] is not y or hit[1] != y._version:
            p = sampling_probabilities(y, self.sampling_rate)
            n_real = getattr(g, "num_real_nodes", None)
            if n_real is not None:
                p[n_real:] = -1.0
            if hit is not None and getattr(g, "_stable_storage", False) and hit[2].shape == p.shape:
                hit[2].copy_(p)                      # same address: what the captured loss kernel reads
                p = hit[2]
            elif getattr(g, "_stable_storage", False):
                g._refresh_hooks.append(lambda: self._arena_rows_check(g, self._sampling(g)))
            store[self.sampling_rate] = (y, y._version, p)
        return store[self.sampling_rate][2]

    def _touched(self) -> None:
        """The parameters are about to change behind torch's back (the fused optimizer kernel writes the flat bucket through raw
        pointers: no version counter moves): advance the model's update epoch, which infer.ForwardRunner reads."""
        m = self.model
        m._spgnn_param_epoch = getattr(m, "_spgnn_param_epoch", 0) + 1

    def step(self, g, draws: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Returns the (global) loss as a device scalar; never synchronises with the host."""
        self._touched()
        return self._back(self._reduce(self._front(g, draws)))

    def _front(self, g, draws: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Mask draw, forward, loss sums, backward; leaves the local gradient sums in the flat bucket (its last slot
        carries the local class-weight sum) and returns the local loss numerator."""
        b = self.bucket
        b.detach_grads()
        y = g.ndata["y"]
        p = self._sampling(g)
        on_gpu = b.flat_param.is_cuda
        pool = ops.scale_pool(b.flat_param.device) if on_gpu else None
        # the dropout seed offset belongs to THIS step object: it is visible to the kernels only while this step's forward
        # and backward are being issued (a captured step leaves no offset behind for other models / eager layers)
        prev_off = ops.DROPOUT_SEED_OFFSET
        ctr = getattr(self, "_seed_ctr", None)
        if pool is not None:
            pool.begin(counter=ctr)          # ONE launch: every GEMM operand's scale block of this step re-armed, the counter advanced
        elif ctr is not None:
            ctr.add_(1)
        draw_seed = 0
        if draws is None:
            if getattr(self, "_use_default_rng", False) and on_gpu and ctr is not None:
                # captured step: the loss kernel draws the mask itself from (seed, step counter, node) - no generator state
                # to restore before a replay, no launch of its own
                draw_seed = self._mask_seed
            else:
                draws = torch.rand(p.shape, device=p.device, generator=self.gen)
        # every GATConv's attention-vector gradient pass is collected during backward and issued as ONE launch after it
        queue = ops.AttnGradQueue(b.flat_param.device) if (on_gpu and ops.DEFER_ATTN_GRADS) else None
        # ... and so are the split-K reductions behind every weight gradient (nothing inside the backward pass reads one)
        sums = ops.StepSums(b.flat_param.device) if (on_gpu and ops.DEFER_STEP_SUMS) else None
        # That is sound only while autograd takes every deferred output over untouched.  The FIRST step of this object proves it
        # for this model (its structure - shared parameters, hooks, retain_grad - is what decides, and is the same on every later
        # step): the deferred outputs start as NaN, so whatever autograd copied, added or handed to a hook before the flush shows
        # up as a NaN in the gathered bucket (one host read, once).  Never under stream capture.
        verify = bool(sums is not None and self._verify_deferred and not torch.cuda.is_current_stream_capturing())
        # the weight-gradient products go to a side stream and overlap the traversals that follow them (ops.SideLaunch)
        side = None
        if on_gpu and ops.OVERLAP_TN:
            if getattr(self, "_tn_side", None) is None:
                self._tn_side = ops.SideLaunch(b.flat_param.device)
            side = self._tn_side
        prev_poison = ops.DEBUG_POISON_DEFERRED
        try:
            if verify:
                ops.DEBUG_POISON_DEFERRED = True
            if ctr is not None:
                ops.DROPOUT_SEED_OFFSET = ctr
            ops.ATTN_GRAD_QUEUE = queue
            ops.STEP_SUMS = sums
            ops.TN_SIDE = side
            rows = None
            if self.loss_rows_only and on_gpu and not getattr(self, "_rows_not_taken", False):
                cap = self._loss_rows_cap(g, p)
                if cap:
                    if self._rows_cnt is None:
                        self._rows_cnt = torch.zeros((2,), dtype=torch.int32, device=p.device)
                    rows = ops.loss_rows(p, draws, draw_seed, cap, cnt=self._rows_cnt,      # the same draw the loss kernel makes
                                         forward=self.loss_rows_only is True)
            ops.LOSS_ROWS = rows
            # the loss itself joins the node that ends in the classifier where that node can take it (ops.LossHead: one pass
            # for logits, loss sums, logit gradient and the classifier's own gradients); dense steps only
            head = None
            if on_gpu and ops.FUSED_LOSS_HEAD and (rows is None or not rows.forward) and (rows is not None or not self.loss_rows_only):
                # (a "backward" loss-rows step has a dense forward: the same kernel, NaN when the step's list overflowed)
                head = ops.LossHead(y, p, draws, draw_seed, self.class_weight, b.sums_slot,
                                    flag=self._rows_cnt if rows is not None else None)
            ops.LOSS_HEAD = head
            logits = self.model(g)[0]
            ops.LOSS_ROWS = None
            ops.LOSS_HEAD = None
            if rows is not None and not rows.used:
                rows = None                  # this model's head does not take the list: its logits have one row per node
                self._rows_not_taken = True  # (a property of the model: later steps do not make the list)
            direct = logits.is_cuda
            if head is not None and head.used:
                # sums already in the bucket's tail; the stored gradient of the numerator starts the backward pass
                num, den = b.loss_slot, b.wsum_slot
                torch.autograd.backward(logits, head.g_logits)
            elif direct:                     # one kernel: mask, log-softmax, weighted NLL sums and the gradient; the two sums
                nd = ops.masked_ce_sums(logits, y, draws, p, self.class_weight, out=b.sums_slot, draw_seed=draw_seed,
                                        unit_grad=True, rows=rows if (rows is not None and rows.forward) else None,
                                        flag=self._rows_cnt if (rows is not None and not rows.forward) else None)   # land in the bucket's tail
                num, den = nd[0], nd[1]
                if self._one is None or self._one.device != num.device:
                    self._one = torch.ones((), dtype=torch.float32, device=num.device)
                torch.autograd.backward(num, self._one)      # gradient exactly 1 (unit_grad), from a tensor that needs no fill
            else:
                num, den = weighted_nll_sums(logits, y, mask_from_draws(draws, p), self.class_weight)
                num.backward()
            if queue is not None:
                queue.flush()                # (inside the step's scale-pool window: its partial sums take no block, but stay in order)
            if side is not None:
                side.join()                  # every weight-gradient product is complete before anything sums its partials
            if sums is not None:
                sums.check_taken_over(b.params)      # every deferred output must BE a parameter's .grad by now
                sums.flush()                 # the attention queue's reductions included: it found this queue installed
        finally:
            ops.DROPOUT_SEED_OFFSET = prev_off
            ops.ATTN_GRAD_QUEUE = None
            ops.STEP_SUMS = None
            ops.TN_SIDE = None
            ops.LOSS_ROWS = None
            ops.LOSS_HEAD = None
            if side is not None:
                side.join()                  # (also on the error path: never leave the side stream dangling in a capture)
            ops.DEBUG_POISON_DEFERRED = prev_poison
            if pool is not None:
                pool.end()
        b.gather_grads()
        if verify:
            if not bool(torch.isfinite(b.flat_grad[:b.numel]).all()):
                raise RuntimeError(
                    "a deferred split-K gradient sum was read before it was filled (a parameter shared by two layers or used twice, "
                    "a gradient hook, retain_grad, a custom backward that reads a weight gradient) - or the first step's gradients "
                    "are not finite for another reason.  Set spgnn_amd.ops.DEFER_STEP_SUMS = False and DEFER_ATTN_GRADS = False for "
                    "this model.")
            self._verify_deferred = False
        if not direct:
            b.wsum_slot.copy_(den.detach().reshape(1))
            b.loss_slot.copy_(num.detach().reshape(1))
        return b.loss_slot

    def _reduce(self, loss_num: torch.Tensor) -> torch.Tensor:
        """The step's only exchange: ONE sum all-reduce of the flat gradient bucket (RCCL); the class-weight sum and the loss
        numerator ride in its last two slots."""
        if self.exchange:
            dist.all_reduce(self.bucket.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
        return loss_num

    def _back(self, loss_num: torch.Tensor) -> torch.Tensor:
        b = self.bucket
        if b.flat_param.is_cuda:             # ONE launch: 1 / weight sum, the fused SGD and the loss scalar
            n = b.numel
            if self._loss_out is None:
                self._loss_out = torch.zeros((1,), dtype=torch.float32, device=b.flat_param.device)
            if self.loss_rows_only and self._skipped is None:
                self._skipped = torch.zeros((1,), dtype=torch.int32, device=b.flat_param.device)
            ops.sgd_momentum_step_(b.flat_param[:n], b.flat_grad[:n], b.flat_mom[:n], self.lr, self.momentum, self.weight_decay,
                                   first_step=(b.steps == 0), lr_dev=self._lr_dev, weight_sum=b.wsum_slot, loss_num=b.loss_slot,
                                   loss_out=self._loss_out, skipped=self._skipped if self.loss_rows_only else None)
            b.steps += 1
            return self._loss_out.reshape(())
        inv = torch.reciprocal(b.wsum_slot)
        self._apply_update(inv)
        b.steps += 1
        return loss_num.reshape(()) * inv[0]

    def _apply_update(self, inv: torch.Tensor) -> None:
        """Fused SGD(momentum) over the flat bucket: one HIP launch (spgnn_sgd_momentum_step)."""
        b = self.bucket
        n = b.numel                          # the parameters only: the bucket's tail slots carry the weight sum and the loss
        ops.sgd_momentum_step_(b.flat_param[:n], b.flat_grad[:n], b.flat_mom[:n], self.lr, self.momentum, self.weight_decay,
                               first_step=(b.steps == 0), grad_scale=inv, lr_dev=self._lr_dev)

    def range_violations(self) -> int:
        """GEMM operands of the steps so far that had rows / blocks outside the split products' 2^18 accuracy envelope
        (ops.range_violations: a device counter; this call synchronises).  0 is the normal state; when it moves, the products'
        wide-range form (``ops.GEMM_WIDE``; ``range_policy="auto"`` switches to it by itself) restores fp32 accuracy for
        operands within ~2^28 of their maximum; ``ops.GEMM_MODE = "fp32"`` (rocBLAS) is the unconditional fallback."""
        return ops.range_violations(self.bucket.flat_param.device) if self.bucket.flat_param.is_cuda else 0

    def set_lr(self, lr: float):
        self.lr = lr
        if self._lr_dev is not None:
            self._lr_dev.fill_(lr)

    # ---- checkpointing: the layout of torch.optim.SGD's state_dict (the reference's ``optimizer_dict``) -----
    def state_dict(self) -> dict:
        """``{"state": {i: {"momentum_buffer": tensor}}, "param_groups": [...]}`` as ``torch.optim.SGD`` over the same
        parameter list writes it (reference job_runner.py:336-343 stores ``optimizer.state_dict()``), so either side can
        load the other's file; plus ``"spgnn"``: the step count, the attention-dropout seed counter and the mask
        generator's state, which a resumed run needs to continue the same random streams."""
        b = self.bucket
        state = {}
        for i, (p, off) in enumerate(zip(b.params, b.offsets)):
            n = p.numel()
            if b.steps > 0:                      # torch creates the buffer at the first step
                state[i] = {"momentum_buffer": b.flat_mom[off:off + n].view_as(p).detach().clone()}
        group = {"lr": self.lr, "momentum": self.momentum, "dampening": 0, "weight_decay": self.weight_decay,
                 "nesterov": False, "maximize": False, "foreach": None, "differentiable": False, "fused": None,
                 "params": list(range(len(b.params)))}
        ctr = getattr(self, "_seed_ctr", None)
        return {"state": state, "param_groups": [group],
                "spgnn": {"steps": b.steps, "seed_ctr": int(ctr.item()) if ctr is not None else 0,
                          "generator": self.gen.get_state().clone(), "mask_seed": self._mask_seed}}

    def load_state_dict(self, sd: dict) -> None:
        """Inverse of :meth:`state_dict`; also takes a plain ``torch.optim.SGD`` state dict (no ``"spgnn"`` entry: the
        step count is then 1 when momentum buffers exist, else 0).  Works on a captured step too: momentum, learning
        rate and counters live in device tensors the graphs read."""
        b = self.bucket
        groups = sd.get("param_groups", [])
        if len(groups) != 1 or len(groups[0].get("params", [])) != len(b.params):
            raise ValueError("optimizer state does not match this model: expected one parameter group of "
                             f"{len(b.params)} parameters")
        g0 = groups[0]
        if g0.get("nesterov") or g0.get("dampening", 0) != 0 or g0.get("maximize"):
            raise ValueError("only plain SGD with momentum is supported (nesterov / dampening / maximize are set)")
        state = sd.get("state", {})
        ids = g0["params"]
        have = 0
        for i, (p, off) in enumerate(zip(b.params, b.offsets)):
            n = p.numel()
            st = state.get(ids[i], state.get(str(ids[i])))
            buf = None if st is None else st.get("momentum_buffer")
            if buf is not None:
                if buf.numel() != n:
                    raise ValueError(f"momentum buffer {i} has {buf.numel()} elements, the parameter {n}")
                b.flat_mom[off:off + n].copy_(buf.reshape(-1).to(b.flat_mom.device, torch.float32))
                have += 1
            else:
                b.flat_mom[off:off + n].zero_()
        self.momentum, self.weight_decay = float(g0.get("momentum", self.momentum)), float(g0.get("weight_decay", self.weight_decay))
        self.set_lr(float(g0.get("lr", self.lr)))
        extra = sd.get("spgnn")
        b.steps = int(extra["steps"]) if extra else (1 if have else 0)
        if have and b.steps == 0:
            b.steps = 1                          # buffers exist: the next step must accumulate into them, not overwrite
        if extra:
            self.gen.set_state(extra["generator"].cpu())
            self._mask_seed = int(extra.get("mask_seed", self._mask_seed))
            ctr = getattr(self, "_seed_ctr", None)
            if ctr is not None:
                ctr.fill_(int(extra["seed_ctr"]))
        # (steps == 0 with zeroed buffers is exact under a captured graph too: the kernel's first-step form equals
        # momentum * 0 + g)

    # ---- HIP-graph replay of the static-graph step ---------------------------------------------------------
    def capture(self, g, warmup: int = 3):
        """Capture one optimizer step on the static batched graph ``g`` into HIP graphs (the reference takes 300
        steps on each batched graph, job_runner.py:1892).  Eagerly the step is ~230 launches of a few microseconds
        each plus autograd's host work, and the host, not the GPU, sets the pace once the kernels are fast enough; a
        replay is two graph launches.  Two graphs, split where the ranks exchange: ``front`` = mask draw, forward, loss,
        backward, gradient gather; [eager: the RCCL all-reduces of ``_reduce``]; ``back`` = SGD update and the loss
        scalar.  One process has nothing to do between the halves and captures both into ONE graph.
        Randomness stays fresh per replay: every mask of the step (node sampling, feature and attention dropout) is a
        counter hash of a host seed frozen at capture plus a device counter (ops.DROPOUT_SEED_OFFSET) that the captured
        step advances in its first launch (spgnn_step_begin) - no generator state to restore before a replay; the
        learning rate is read from a device scalar (``set_lr`` keeps working)."""
        dev = self.bucket.flat_param.device
        if self._lr_dev is None:                   # created once: earlier captures (other batch arenas) keep reading them
            self._lr_dev = torch.full((1,), float(self.lr), dtype=torch.float32, device=dev)
            self._seed_ctr = torch.zeros(1, dtype=torch.int64, device=dev)     # installed as ops.DROPOUT_SEED_OFFSET inside _front only
        self._use_default_rng = True
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):        # warm-up on the capture stream: lazy inits, allocator pools, steps > 0
                self._static_loss = self.step(g)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # thread-local capture mode: other threads (the process group's watchdog) may touch the runtime meanwhile
        # whatever persistent buffer the recorded launches address outside the graphs' own pool (the weight-prep operand sets
        # of ops / ops_bf16) is collected here and lives as long as this step can replay
        prev_refs, ops.CAPTURE_REFS = ops.CAPTURE_REFS, []
        try:
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                self._front(g)
                if not self.exchange and ONE_GRAPH_PER_STEP:     # nothing happens between the halves: one graph launch per step
                    self._static_loss = self._back(self.bucket.loss_slot)
            self._graph_back = None
            if self.exchange or not ONE_GRAPH_PER_STEP:
                self._graph_back = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph_back, capture_error_mode="thread_local"):
                    self._static_loss = self._back(self.bucket.loss_slot)
            self._capture_refs = ops.CAPTURE_REFS
        finally:
            ops.CAPTURE_REFS = prev_refs
        self._captured_graph = g                    # the node data and index arrays the graphs read
        # kept per ARENA graph (a handful of size classes, each reused by every later batch of its class); a capture on an ordinary
        # graph replaces the previous such capture - the r3 pattern "capture every loader batch" must not pile up graphs and their
        # private memory pools
        key = id(g) if getattr(g, "_stable_storage", False) else "adhoc"
        self._captures[key] = {"graph": g, "front": self._graph, "back": self._graph_back, "loss": self._static_loss,
                               "refs": self._capture_refs}
        self.capture_steps = max(warmup, 1)        # optimizer steps the warm-up took on ``g`` (they count towards GCN_STEPS)
        return self

    def select(self, g) -> bool:
        """Make the capture recorded on graph ``g`` the one :meth:`replay` runs (a step object holds one capture per batch
        arena).  -> whether there is one."""
        rec = self._captures.get(id(g) if getattr(g, "_stable_storage", False) else "adhoc")
        if rec is None or rec["graph"] is not g:
            return False
        self._graph, self._graph_back, self._static_loss, self._capture_refs = rec["front"], rec["back"], rec["loss"], rec["refs"]
        self._captured_graph = g
        return True

    def replay(self) -> torch.Tensor:
        if self._graph is None:
            raise RuntimeError("no captured step is selected (capture() first; the selected capture's arena may have been evicted)")
        self._touched()
        self._graph.replay()
        if self._graph_back is not None:
            self._reduce(self.bucket.loss_slot)
            self._graph_back.replay()
        return self._static_loss

    # ---- the reference's loader-batch cycle (job_runner.py:1870-1920): GCN_STEPS steps on every freshly built batch ----
    def arena_graph(self, g, granule: int = 256):
        """``g`` (a device batch, e.g. data.assemble_batch) copied into the batch arena of its size class - created on first
        use - and padded to the class (spgnn_amd/arena.py).  -> the arena's graph, the object to capture / replay on."""
        from .arena import BatchArena
        key = BatchArena.class_key(g, granule)
        arena = self._arenas.pop(key, None)
        if arena is None:
            while len(self._arenas) >= self.max_arenas:          # least recently used class goes, with its capture
                _, old = next(iter(self._arenas.items()))
                self._captures.pop(id(old.graph), None)
                if self._captured_graph is old.graph:            # never leave replay() pointing at graphs whose buffers are freed
                    self._graph = self._graph_back = self._captured_graph = None
                del self._arenas[next(iter(self._arenas))]
            arena = BatchArena(g, granule)
        self._arenas[key] = arena                                # re-inserted last: dict order is the recency order
        return arena.load(g)

    def run_batches(self, batches, steps: int, assemble, granule: int = 256):
        """The whole loader loop of job_runner.py:1870-1920: for every host batch ``b`` of ``batches`` - ``g = assemble(b)`` (e.g.
        ``lambda b: data.assemble_batch(b, "cuda", POS_ENC_DIM)``), then ``steps`` optimizer steps on it - with the assembly of
        batch i + 1 (pinned packing, uploads, device CSC, anchors, distance encoding: ~8 ms at 64 trees) issued on a side stream
        right after batch i's replays were queued, so it runs UNDER them; between two batches only the arena load (~1 ms) is
        left.  Same arithmetic as ``run_batch`` per batch.  -> the last-step losses (device scalars cloned per batch)."""
        dev = self.bucket.flat_param.device
        if dev.type != "cuda":
            raise RuntimeError("run_batches replays captured HIP graphs: it needs a ROCm device")
        main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)
        it = iter(batches)

        def stage(b, after):
            if after is not None:
                side.wait_event(after)           # the previous staged batch was copied into its arena: its blocks may be reused
            with torch.cuda.stream(side):
                g = assemble(b)
            ev = torch.cuda.Event()
            ev.record(side)
            return g, ev

        losses, loaded = [], None
        first = next(it, None)
        nxt = stage(first, None) if first is not None else None
        while nxt is not None:
            g, ready = nxt
            main.wait_event(ready)
            self._range_policy_check()
            ag = self.arena_graph(g, granule)    # copies on the main stream, ordered after the assembly
            loaded = torch.cuda.Event()
            loaded.record(main)
            del g
            loss = self._replays_on(ag, steps)   # queued, not waited for
            b = next(it, None)
            nxt = stage(b, loaded) if b is not None else None      # ... and the next assembly runs under them
            losses.append(loss.clone())
        return losses

    def _range_policy_check(self) -> None:
        if self.loss_rows_only:
            self.check_loss_rows()                   # (one 4-byte read per loader batch: the flag of the previous batch's steps)
        if self.range_policy == "auto" and not ops.GEMM_WIDE and self.bucket.flat_param.is_cuda:
            v = self.range_violations()              # one 4-byte read per loader batch (the flags of the previous batch's steps)
            if self._violations_seen is not None and v > self._violations_seen:
                ops.GEMM_WIDE = True                 # wide-range products from here on; the narrow captures and images are stale
                self._captures.clear()
                self._graph = self._graph_back = None
            self._violations_seen = v

    def _replays_on(self, ag, steps: int) -> torch.Tensor:
        done = 0
        if not self.select(ag):
            # the warm-up steps of a capture are real optimizer steps: never more of them than the caller asked for
            # (GCN_STEPS < 3 would otherwise over-train the first batch of a class), and with several ranks every rank must
            # issue exactly ``steps`` all-reduces for this batch whether it captures or replays
            if steps < 1:
                raise ValueError("run_batch / run_batches need steps >= 1 (a capture takes at least one optimizer step)")
            self.capture(ag, warmup=min(3, steps))
            done = self.capture_steps
        loss = self._static_loss                     # the device scalar every replay of this capture writes
        for _ in range(max(steps - done, 0)):
            loss = self.replay()
        return loss

    def run_batch(self, g, steps: int, granule: int = 256) -> torch.Tensor:
        """``steps`` optimizer steps on loader batch ``g`` as HIP-graph replays: the first batch of a size class pays the
        warm-up steps and the capture, every later one only the copies into the arena.  -> the last step's loss (device)."""
        self._range_policy_check()
        return self._replays_on(self.arena_graph(g, granule), steps)
