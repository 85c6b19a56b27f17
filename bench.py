#!/usr/bin/env python3
"""bench.py — message-passing edges/sec (fwd+bwd) of the GNN training step on N MI355X.

One "step" = one optimizer step on one static batched graph of synthetic random-fan-out trees
(SURVEY.md Appendix D): Bernoulli node mask -> model forward -> class-weighted CE -> backward ->
[RCCL all-reduce of the flat gradient bucket] -> fused SGD(momentum).  Nothing is skipped inside
the timed region.  Default workload: st_pgat_spgnn_3 (full SPGNN + LSPE position stream), 512 trees
per GPU, fp32, dropout on — the configuration BASELINE.json's north_star quotes its target on; with
N GPUs every rank gets its own 512 trees (weak scaling, global batch 512*N).
`--config st_gat_6 --dtype bf16` is BASELINE config 4 (bf16 storage, fp32 accumulate).

Single GPU:  python bench.py [--steps K --warmup W]
Multi GPU:   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                 --master-port P bench.py --gpus N --steps K --warmup W
        or:  python bench.py --gpus N --steps K --warmup W      (no launcher: bench.py starts the N ranks itself, self_launch)

Rank 0 prints ONE JSON line (contract in the task statement).  Protocol: W warm-up steps, then
exactly K timed steps between barrier + synchronize on both sides (`value`, `ms_per_step`: max over ranks); every
timed step is also bracketed by HIP events on the compute stream (`step_ms`: median / p10 / p90).  Extra objects:
  roofline        the step's dominant kernel, FIXED per dtype (f32: spgnn_gemm_nt, all its launches of a step, MFMA bound;
                  bf16: the GAT message-passing kernels spgnn_gat_{fwd,bwd_dst,bwd_src}_bf16, HBM bound): algorithmic
                  flops or bytes per step / its measured time per step (HIP events on the launch stream) / peak
  roofline_k123   SURVEY.md §8d's K1-K3 (GAT kernels): vs the survey's per-model byte count and vs their own bytes
  composite       SURVEY.md §8d / BASELINE.md §2: t_graph + max(t_gemm_bytes, t_gemm_flops) against the measured step
  copy_bandwidth  achievable HBM bandwidth of a plain device copy on this box, measured in the same run
  cpu_baseline    the DGL-CPU-equivalent restatement (oracle/, "port") timed on this host's cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
MFMA_F16_PEAK = 2500.0      # dense fp16 / bf16 TFLOP/s
FP32_MATRIX_PEAK = 157.3    # native fp32 MFMA TFLOP/s (the pipe an fp32 GEMM would otherwise use)
GEMM_NAMES = ("gemm_nt", "gemm_tn", "gemm_nt_pair", "gemm_tn_pair", "gemm_nt_bf16", "gemm_tn_bf16", "gemm_nt_skinny")
HELPER_NAMES = ("absmax", "split_rows")
GAT_PREFIXES = ("gat_fwd", "gat_bwd", "gat_agg", "lspe_")
# roofline.traffic is NOT measured by this run: PMC counters need rocprofv3 passes of their own
TRAFFIC_SOURCE = "profiles/traffic_latest.json (builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this step, per launch; not measured in this run)"


def algorithmic_bytes(key) -> float:
    """Algorithmic HBM bytes of ONE launch (SURVEY.md §8d per-unit figures): every operand row
    read once, every result row written once, index arrays once.  s = bytes per stored row element (4, or 2 for the
    *_bf16 kernels; scores, attention and indices are always 4 bytes)."""
    name = key[0]
    s = 2 if name.endswith("_bf16") else 4
    base = name[:-5] if name.endswith("_bf16") else name
    if base == "gat_fwd":            # read ft [+res]; write out [and/or its fp32 head mean]; read el, er; write a; CSC
        _, N, E, H, D, has_res, mean, has_out = key
        return (s * (1 + has_res + has_out) * N * H * D + 4 * mean * N * D + 4 * (2 * N * H + E * H)
                + 4 * (N + 1 + E))
    if base == "gat_bwd_dst":        # read g_out [,out], ft; write g_pre; read el, er, a; write g_e, g_er; CSC
        _, N, E, H, D, act, mean = key
        return (s * (2 + (1 if act else 0)) * N * H * D + (4 * N * D if mean else s * N * H * D)
                + 4 * (3 * N * H + 2 * E * H) + 4 * (N + 1 + E))
    if base == "gat_bwd_src":        # read g_pre, write g_ft; read a, g_e; write g_el; CSR + slot map
        _, N, E, H, D = key
        return s * 2 * N * H * D + 4 * (N * H + 2 * E * H) + 4 * (N + 1 + 2 * E)
    if base == "lspe_fwd":           # one level, three heads of D: read ft + res (3D each), write the next structure input (3D)
        _, N, E, D = key             # and the next position input (D); el, er in, a out (3 heads); CSC
        return 4 * 10 * N * D + 4 * (2 * 3 * N + 3 * E) + 4 * (N + 1 + E)
    if base == "lspe_bwd_dst":       # read g_buf, out (3D each), [g_xp,] xp (D each), gather ft (3D); write g_pre (3D);
        _, N, E, D, has_g2 = key     # el, er, a in; g_e, g_er out; CSC
        return 4 * (13 + has_g2) * N * D + 4 * (3 * 3 * N + 2 * 3 * E) + 4 * (N + 1 + E)
    if base == "lspe_bwd_src":       # gather g_pre (3D), write g_ft (3D); a, g_e, g_er in; g_el out; CSR + slot map
        _, N, E, D = key
        return 4 * 6 * N * D + 4 * (2 * 3 * N + 2 * 3 * E) + 4 * (N + 1 + 2 * E)
    if base == "scores_fwd":         # read x; write S
        _, N, K, J = key
        return s * N * K + 4 * N * J
    if base == "scores_bwd_w":       # read x, gS; partials negligible
        _, N, K, J = key
        return s * N * K + 4 * N * J
    if base == "scores_bwd_w_pair":  # two such passes in one launch
        _, N, K0, J0, K1, J1 = key
        return s * N * (K0 + K1) + 4 * N * (J0 + J1)
    if base == "scores_bwd_w_multi": # every layer's pass in one launch: key = (N, sum K, sum J, jobs, bytes per row element)
        _, N, Ks, Js, _n, sb = key
        return sb * N * Ks + 4 * N * Js
    if base == "scores_bwd_x":       # read + write gX; read gS
        _, N, K, J = key
        return 4 * 2 * N * K + 4 * N * J
    if base == "gat_agg_fwd":        # read x; write the z blocks [with a copy of x per head]; el, er, a; CSC
        _, N, E, H, F_, xcopy = key      # xcopy: 0 none, 1 a copy of x in every head's block, 2 one copy behind the last block
        return s * N * F_ + s * N * F_ * (2 * H if xcopy == 1 else H + 1 if xcopy == 2 else H) + 4 * (2 * N * H + E * H) + 4 * (N + 1 + E)
    if base == "gat_agg_bwd_dst":    # read the z part of g_z and x; el, er, a; write g_e, g_er; CSC
        _, N, E, H, F_ = key
        return s * N * (H + 1) * F_ + 4 * (3 * N * H + 2 * E * H) + 4 * (N + 1 + E)
    if base == "gat_agg_bwd_src":    # read g_z (both parts); write g_x; a, g_e, g_er, g_el; CSR + slot map
        _, N, E, H, F_ = key
        return s * N * (2 * H + 1) * F_ + 4 * (2 * N * H + 2 * E * H) + 4 * (N + 1 + 2 * E)
    if base == "head_mean":
        _, N, H, D = key
        return 4 * N * (H + 1) * D
    if base == "act_bwd":            # read g_out [and out]; write g_pre
        _, N, H, D, act, mean = key
        return 4 * N * (D if mean else H * D) + 4 * N * H * D * (2 if act else 1)
    if base == "act_bwd_proj":       # read g_logits and out; write g_pre (the classifier's input gradient stays in registers)
        _, N, H, D, act, J = key
        return 4 * N * H * D * (2 if act else 1) + 4 * N * J + 4 * J * D
    if base == "sum_partials":
        return 0.0
    if base == "scores_from_parts":      # read the (N, H*D/64, 2) partials, write (N, 2H)
        _, N, H, D = key
        return 4 * 2 * N * H * (D // 64) + 4 * 2 * N * H
    if base == "masked_ce":
        _, N, C = key
        return 4 * 2 * N * C + 4 * 4 * N
    if base == "classifier_ce":      # classifier + loss + its weight gradient from one read of the rows: x in, logits and g_logits out
        _, N, K, J = key
        return s * N * K + 4 * 2 * N * J
    if base == "loss_rows":          # the step's mask as a row list (TrainStep(loss_rows_only=True)): read p twice, write idx and inv
        _, N, cap = key
        return 4 * 3 * N + 4 * cap
    if base == "gather_rows":        # listed rows in, list-order rows out
        _, cap, C = key
        return 4 * 2 * cap * C
    if base == "expand_rows":        # list-order rows in (at most), node-order rows out
        _, N, C = key
        return 4 * N * C
    if base in ("gemm_nt", "gemm_tn", "gemm_nt_pair", "gemm_tn_pair") or base in HELPER_NAMES:
        return 0.0                    # compute-side kernels: accounted in "gemm" / "composite", not in the message-passing bytes
    if base == "spmm_sum":
        _, N, E, F = key
        return 4 * 2 * N * F + 4 * (N + 1 + E)
    if base in ("spmm_max_fwd", "spmm_max_bwd"):      # rows in + rows out + the argmax (one byte per element in the compact form)
        _, N, E, F = key
        q = F // 4
        compact = F % 4 == 0 and any(q % t == 0 and q // t in (1, 2, 4, 8) for t in (64, 32, 16))     # spgnn_spmm_max_u8_supported
        return 4 * 2 * N * F + (1 if compact else 4) * N * F + 4 * (N + 1 + (E if base == "spmm_max_fwd" else 2 * E))
    raise KeyError(name)


def gemm_bytes(key) -> float:
    """Algorithmic HBM bytes of one projection product: both operands once + the result once.  A pair launch
    (``gemm_nt_pair`` / ``gemm_tn_pair``: two independent products, key = both shapes) is the sum of its two products."""
    name = key[0]
    if name.endswith("_pair"):
        return gemm_bytes((name[:-5],) + tuple(key[1:4])) + gemm_bytes((name[:-5],) + tuple(key[4:7]))
    _, a, b, c = key
    s = 2 if name.endswith("_bf16") else 4
    if name.startswith("gemm_nt"):           # (M, N, K): A (M,K), B (N,K) -> C (M,N) in the storage dtype
        return s * (a * c + b * c + a * b)
    return s * (a * b + a * c) + 4 * b * c   # gemm_tn (R, M, N): A (R,M), B (R,N) -> fp32 C (M,N)


def gemm_flops(key) -> float:
    """Algorithmic flops (2 M N K of the fp32 product) of one launch."""
    if key[0].endswith("_pair"):
        return 2.0 * key[1] * key[2] * key[3] + 2.0 * key[4] * key[5] * key[6]
    return 2.0 * key[1] * key[2] * key[3]


def survey_k123_bytes(gat_keys, dtype_bytes) -> float:
    """SURVEY.md §8d's per-GATConv figure B_fwd + B_bwd (= 5.29 GB for st_pgat_spgnn_3 at 512 trees in fp32) from the
    layer shapes that ran: B_fwd = s 2 N HD + 4 (2 N H + E H) + 4 (N + 1 + E); B_bwd = s 3 N HD + 4 (2 N H + 3 E H) +
    8 (N + 1 + E).  ``gat_keys``: [((name, N, E, H, D), launches per step)], one entry per GATConv layer."""
    tot, s = 0.0, dtype_bytes
    for (_, N, E, H, D), n in gat_keys:
        tot += n * (s * 5 * N * H * D + 4 * (4 * N * H + 4 * E * H) + 12 * (N + 1 + E))
    return tot


def usable_cores(cap: int = 32) -> int:
    """Threads the baseline may really use: scheduler affinity and cgroup quota, not os.cpu_count()
    (the GPU box reports 256 logical CPUs but grants far fewer; oversubscribing made one iteration 50 s)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def _cpu_sample(cfg, model, samples, n_trees, iters, warm, budget_s):
    """One timed sample of the DGL-CPU-equivalent restatement: fwd + loss + bwd on the first ``n_trees`` trees."""
    from oracle import dgl_cpu as O
    from spgnn_amd import synthetic
    from spgnn_amd.configs import class_weight_list
    cores = usable_cores()
    torch.set_num_threads(cores)
    g = synthetic.batch_from_samples(samples[:n_trees], "cpu", cfg.POS_ENC_DIM)
    src, dst = g.edges()
    n, E = g.number_of_nodes(), g.number_of_edges()
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS))
    y = g.ndata["y"]
    mask = torch.rand(n) < torch.where(y != 0, torch.tensor(1.0), torch.tensor(cfg.SAMPLING_RATE))
    times = []
    t_start = time.perf_counter()
    for i in range(warm + iters):
        if len(times) >= min(2, iters) and time.perf_counter() - t_start > budget_s:
            break
        t0 = time.perf_counter()
        out = O.net_forward(cfg.KIND, sd, src, dst, n, g.ndata["fvs"], g.ndata.get("pos_enc"))[0]
        loss = O.masked_weighted_ce(out, y, mask, w)
        grads = torch.autograd.grad(loss, [p for p in sd.values() if p.requires_grad], allow_unused=True)
        del grads, out, loss
        if i >= warm:
            times.append(time.perf_counter() - t0)
    import statistics
    med = statistics.median(times)
    return {"value": E * cfg.CONV_LAYERS / med, "unit": "layer-edges/s", "cores": cores, "kind": "port",
            "sample": f"{n_trees} trees (N={n}, E={E}) fp32 fwd+bwd, median of {len(times)} after {warm} warm-up, {cores} threads, "
                      f"{med * 1e3:.0f} ms/iter, restated DGL-CPU",
            "ms_per_iter": med * 1e3, "nodes": n, "edges": E, "iters": len(times), "warmups": warm}


def cpu_baseline(cfg, model, samples, n_trees, small_trees=64, small_sample=False):
    """DGL-CPU-equivalent (restated) fwd+bwd on the host cores: oracle/dgl_cpu.py, same weights, eval-mode arithmetic (no
    dropout), fp32, all usable host threads.  The reported sample is the HEADLINE workload itself (all ``n_trees`` trees of
    rank 0's batch: 2 iterations after 1 warm-up, ~7 s each: ~20 s of CPU work); ``small_sample`` adds the 64-tree sample
    of earlier rounds (median of 10 after 3 warm-ups) as ``sample64`` / ``value64``."""
    full = _cpu_sample(cfg, model, samples, n_trees, iters=2, warm=1, budget_s=40.0)
    if small_sample and n_trees > small_trees:
        try:
            small = _cpu_sample(cfg, model, samples, small_trees, iters=10, warm=3, budget_s=25.0)
            full["value64"] = small["value"]
            full["ms_per_iter64"] = small["ms_per_iter"]
            full["sample64"] = small["sample"]
        except Exception as e:          # never lose the headline sample to the small one
            full["sample64"] = "failed: " + repr(e)[:200]
    return full


def copy_bandwidth(dev, mib=1024, iters=10):
    """Achievable HBM bandwidth of a plain device-to-device copy (read + write), median of ``iters``."""
    n = mib * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record()
        torch.cuda.synchronize(dev)
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    med = ts[len(ts) // 2]
    del a, b
    return {"GBps": 2.0 * n * 4 / (med * 1e-3) / 1e9, "frac_of_hbm_peak": 2.0 * n * 4 / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "what": f"torch device copy of {mib} MiB fp32 (read + write), median of {iters}"}


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, max(0, int(round(q * (len(xs) - 1)))))]


def run_leg(config, dtype, trees, steps, warmup, *, rank, world, dev, eager=False, no_eager_leg=False, no_dropout=False,
            no_kernel_timers=False, copy_bw=None, heads=0, loss_rows=False, exchange=None):
    """One measured workload: build the model and the batch, warm up, capture, time ``steps`` steps between barrier +
    synchronize, then the instrumented eager leg.  -> (the JSON object on rank 0 else None, (cfg, model, samples))."""
    from spgnn_amd import _capi, models, ops, synthetic
    from spgnn_amd.configs import class_weight_list, get_config
    from spgnn_amd.train import TrainStep
    _capi.load()                                              # fail loudly if the HIP library is missing
    exchange = world > 1 if exchange is None else bool(exchange)   # ranks exchange gradients (also forced on with ONE rank)

    cfg = get_config(config)
    if heads:                                                 # BASELINE.json words config 2 as "8-head"; the reference file has 2
        cfg.MODEL["num_heads"] = heads
    torch.manual_seed(0)                                      # identical replicas on every rank
    model = models.build_model(cfg.MODEL).to(dev)
    model.init(None)
    model.set_gcn_only()
    model.train(not no_dropout)
    bf16 = dtype == "bf16"
    if bf16:
        models.set_storage_dtype(model, torch.bfloat16)
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)

    samples = synthetic.synthetic_trees(trees, rank=rank)
    g = synthetic.batch_from_samples(samples, dev, cfg.POS_ENC_DIM)
    g.csc(dev)                                                # CSC/CSR built once per loader batch (static for all steps)
    N, E = g.number_of_nodes(), g.number_of_edges()
    step = TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, cfg.OPTIMIZER["lr"],
                     cfg.OPTIMIZER["momentum"], seed=1234 + rank, loss_rows_only=loss_rows, always_exchange=exchange)

    def sync():
        torch.cuda.synchronize()
        if exchange:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up.  Its last two steps are fully instrumented (HIP events around every hand-written kernel launch): they give
    # the per-kernel breakdown.  Recording ~140 event pairs per step perturbs the step, so they are never inside the
    # timed region; the roofline kernels are re-timed on their own (few events per step) after it.
    probe = 0 if no_kernel_timers else min(2, warmup)
    for _ in range(warmup - probe):
        loss = step.step(g)
    kt_all = {}
    overlap_tn = ops.OVERLAP_TN          # the per-kernel legs time every launch ALONE (no weight-gradient product on the side stream)
    if probe:
        sync()
        ops.OVERLAP_TN = False
        ops.KernelTimer.start()
        for _ in range(probe):
            loss = step.step(g)
        kt_all = ops.KernelTimer.stop()
        ops.OVERLAP_TN = overlap_tn
    # The roofline kernel is FIXED per dtype (not "whichever shape won this run"): all launches of it in a step.
    spmm_model = cfg.KIND in ("gcn", "gin", "sage")           # rows D / E / F: the SpMM kernels are the roofline kernel (HBM-bound)
    roof_names = (("gat_fwd_bf16", "gat_bwd_dst_bf16", "gat_bwd_src_bf16", "gat_agg_fwd_bf16", "gat_agg_bwd_dst_bf16",
                   "gat_agg_bwd_src_bf16") if bf16 else ("spmm_sum", "spmm_max_fwd", "spmm_max_bwd") if spmm_model else ("gemm_nt", "gemm_nt_pair"))
    gat_names = tuple(k for k in {k[0] for k in kt_all} if k.startswith(GAT_PREFIXES))
    bracket = [k for k in kt_all if k[0] in roof_names or k[0] in gat_names or k[0] in GEMM_NAMES]
    sync()
    # Timed region.  Default: every step is a replay of the captured step (one HIP graph; two around the gradient
    # all-reduce, TrainStep.capture) - eagerly the launches and autograd's host work per step take the host as long as
    # the GPU needs for the kernels, so a slow host core would be what is measured.  --eager times eagerly issued steps.
    launch, capture_error = "eager", None
    if not eager:
        try:
            step.capture(g)
            launch = "hip-graph replay"
        except Exception as e:                      # never lose the bench line to the capture
            capture_error = repr(e)[:300]
    if world > 1:                                   # every rank must time the same kind of step
        ok = torch.tensor([1.0 if launch != "eager" else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok) == 0.0:
            launch = "eager"
    run_step = step.replay if launch != "eager" else (lambda: step.step(g))
    import gc
    gc.collect()                                    # a collection of the previous leg's objects inside a 20-step leg showed up
    gc.disable()                                    # as 1.54 ms wall against 1.11 ms by events (r03): keep the host out of it.
    # (collected HERE, before the last warm replays: a collection right in front of the timed region left the GPU idle for
    # ~0.1 s and the first timed steps ran at a lower clock - 5.12 ms per step over 20 steps against 4.9-5.0 steady)
    for _ in range(3):                              # replays of the fresh graphs before the clock starts
        loss = run_step()
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    comm_ev = None
    if exchange and launch != "eager":
        # HIP events on the compute stream around the step's one collective (TrainStep._reduce, issued eagerly between the two
        # graphs): what the all-reduce costs inside the replayed step, per step
        comm_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        plain_reduce = step._reduce
        it = iter(comm_ev)

        def timed_reduce(loss_num):
            pair = next(it, None)
            if pair is None:
                return plain_reduce(loss_num)
            pair[0].record()
            r = plain_reduce(loss_num)
            pair[1].record()
            return r
    try:
        if comm_ev is not None:
            step._reduce = timed_reduce
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            loss = run_step()
            marks[i + 1].record()
        sync()
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()
        if comm_ev is not None:
            step._reduce = plain_reduce
    comm = None
    if exchange:
        ar_ms = sorted(a.elapsed_time(b) for a, b in comm_ev) if comm_ev is not None else []
        backend = dist.get_backend()
        try:
            lib_ver = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:
            lib_ver = None
        # the world size as the collective library itself sees it: a sum of ones over the group
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        comm = {"backend": backend + (" (RCCL)" if backend == "nccl" else ""), "library_version": lib_ver,
                "world_size": dist.get_world_size(), "ranks_counted_by_allreduce": int(ones.item()),
                "collective": "one sum all-reduce of the flat fp32 gradient bucket per step (loss normaliser in its tail)",
                "bucket_bytes": int(step.bucket.flat_grad.numel() * 4),
                "allreduce_ms": ({"p50": pct(ar_ms, 0.5), "p90": pct(ar_ms, 0.9), "n": len(ar_ms),
                                  "how": "HIP events on the compute stream around dist.all_reduce inside the timed replays (rank 0)"}
                                 if ar_ms else None)}
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    kt_dom, eager_leg, in_step = {}, None, None
    if not no_eager_leg and not no_kernel_timers:
        # a replay cannot carry HIP events per launch: the roofline kernels' launches are bracketed in eagerly issued
        # steps right after the timed region (same kernels, same operands, same stream; only these kernels carry events)
        n_leg = min(steps, 20)
        ops.OVERLAP_TN = False
        ops.KernelTimer.start(only=bracket or None)
        t1 = time.perf_counter()
        for _ in range(n_leg):
            step.step(g)
        sync()
        eager_leg = {"ms_per_step": (time.perf_counter() - t1) / n_leg * 1e3, "steps": n_leg}
        kt_dom = ops.KernelTimer.stop()
        ops.OVERLAP_TN = overlap_tn
        if not kt_all:
            kt_all, probe = dict(kt_dom), n_leg
        # the same K1-K3 launches as the SHIPPED step runs them: weight-gradient products on the side stream beside them
        side_on = bool(overlap_tn and N >= ops.OVERLAP_TN_MIN_ROWS)
        gat_bracket = [k for k in bracket if k[0] in gat_names]
        if side_on and gat_bracket:
            ops.KernelTimer.start(only=gat_bracket)
            for _ in range(n_leg):
                step.step(g)
            sync()
            kt_in_step = ops.KernelTimer.stop()
            in_step = {"ms_per_step": sum(sum(v) for v in kt_in_step.values()) / n_leg, "steps": n_leg, "side_stream": True}
        elif gat_bracket:       # no side stream in this configuration: the step runs these launches as the isolated leg does
            in_step = {"ms_per_step": sum(sum(kt_dom[k]) for k in gat_bracket if k in kt_dom) / n_leg, "steps": n_leg, "side_stream": False}
    loss_val = float(loss)
    if copy_bw is None and rank == 0:
        copy_bw = copy_bandwidth(dev)

    tot = torch.tensor([elapsed, float(E), float(N)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, E_all, N_all = float(mx[0]), float(sm[1]), float(sm[2])
    else:
        E_all, N_all = float(E), float(N)

    if rank == 0:
        L = cfg.CONV_LAYERS
        ms = elapsed / steps * 1e3
        value = E_all * L * steps / elapsed
        s_row = 2 if bf16 else 4
        gemm_desc = ("bf16 MFMA, single product, fp32 accumulate (bf16 storage)" if bf16 else
                     "split-fp16 x3 MFMA, fp32 accumulate (fp32-GEMM accuracy)" if ops.GEMM_MODE == "f16x3"
                     else "fp32 (rocBLAS/hipBLASLt via torch.mm)")
        out = {
            "metric": "message-passing edges/sec (fwd+bwd), batched trees", "value": value, "unit": "layer-edges/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            # <= 118 characters (the driver's record cuts strings at 120)
            "config": {"workload": f"{config}{f' {heads} heads' if heads else ''} train step fwd+loss+bwd+allreduce+SGD, {trees} trees/GPU n~U[120,180], "
                                   f"{'bf16 rows' if bf16 else 'fp32'}, dropout {'off' if no_dropout else 'on'}"
                                   + ((", loss rows (bwd)" if loss_rows == "backward" else ", loss rows") if loss_rows else ""),
                       "trees_per_gpu": trees, "global_trees": trees * world, "nodes": int(N_all),
                       "edges": int(E_all), "conv_layers": L, "trainable_params": n_params,
                       "parallelism": f"dp{world}", "launch": launch, "gemm": gemm_desc,
                       "weight_gradients_on_side_stream": bool(ops.OVERLAP_TN and N >= ops.OVERLAP_TN_MIN_ROWS)},
            "graph_edges_per_s": E_all * steps / elapsed, "loss": loss_val,
            "step_ms": {"median": pct(step_ms, 0.5), "p10": pct(step_ms, 0.1), "p90": pct(step_ms, 0.9),
                        "n": len(step_ms), "how": "HIP events on the compute stream around every timed step (rank 0)"},
            "copy_bandwidth": copy_bw,
        }
        # The driver's parsed record keeps SCALAR members of `config` / `roofline` / `cpu_baseline` only (nested objects are
        # dropped): everything a reader of that record needs is therefore repeated as flat scalars (flatten_for_driver).
        out["config"]["comm_world_size"] = world
        out["config"]["comm_backend"] = "none (single rank)"
        if comm is not None:
            out["comm"] = comm
            out["config"]["comm_backend"] = comm["backend"]
            out["config"]["comm_library_version"] = comm["library_version"]
            out["config"]["comm_world_size"] = comm["ranks_counted_by_allreduce"]
            out["config"]["comm_bucket_bytes"] = comm["bucket_bytes"]
            if comm["allreduce_ms"]:
                out["config"]["allreduce_ms_p50"] = comm["allreduce_ms"]["p50"]
                out["config"]["allreduce_ms_p90"] = comm["allreduce_ms"]["p90"]
        if kt_all:
            try:                                    # never lose the bench line to the accounting of an unknown kernel key
                nprobe = max(probe, 1)
                n_leg = eager_leg["steps"] if eager_leg else nprobe

                def per_step(key):
                    """(ms per step, launches per step) of one (kernel, shape): from the dedicated leg when it was bracketed there."""
                    if key in kt_dom:
                        return sum(kt_dom[key]) / n_leg, len(kt_dom[key]) / n_leg
                    return sum(kt_all[key]) / nprobe, len(kt_all[key]) / nprobe

                where = ("eagerly issued steps right after the timed region, HIP events on the launch stream around the launches "
                         "of the roofline / GAT / GEMM kernels only") if kt_dom else f"{nprobe} fully instrumented warm-up step(s)"
                traffic_tab = {}
                tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
                if os.path.exists(tpath):
                    try:
                        traffic_tab = json.load(open(tpath))
                    except Exception:
                        traffic_tab = {}

                def traffic_of(keys):
                    """PMC HBM bytes per step of these (kernel, shape) keys, from the committed rocprofv3 --pmc pass, or None."""
                    tot_, seen = 0.0, False
                    for k in keys:
                        t = traffic_tab.get("_".join(str(x) for x in k))
                        if t is None:
                            return None
                        tot_ += t * per_step(k)[1]
                        seen = True
                    return tot_ if seen else None

                # ---- projection GEMMs ---------------------------------------------------------------------------------
                gemm_keys = [k for k in kt_all if k[0] in GEMM_NAMES]
                products = 1 if bf16 else 3
                g_ms = g_fl = g_by = 0.0
                per_name = {}
                for k in gemm_keys:
                    t, n = per_step(k)
                    fl = gemm_flops(k) * n
                    g_ms += t; g_fl += fl; g_by += gemm_bytes(k) * n
                    d = per_name.setdefault(k[0], [0.0, 0.0, 0.0, 0])
                    d[0] += t; d[1] += fl; d[2] += gemm_bytes(k) * n; d[3] += n
                if gemm_keys:
                    out["gemm"] = {
                        "kernels": {nm: {"ms_per_step": d[0], "launches_per_step": d[3], "algorithmic_TFLOPs": d[1] / (d[0] * 1e-3) / 1e12,
                                         "executed_mfma_TFLOPs": products * d[1] / (d[0] * 1e-3) / 1e12,
                                         "algorithmic_GBps": d[2] / (d[0] * 1e-3) / 1e9}
                                    for nm, d in per_name.items()},
                        "ms_per_step": g_ms, "algorithmic_TFLOP_per_step": g_fl / 1e12, "algorithmic_GB_per_step": g_by / 1e9,
                        "algorithmic_TFLOPs": g_fl / (g_ms * 1e-3) / 1e12, "executed_mfma_TFLOPs": products * g_fl / (g_ms * 1e-3) / 1e12,
                        "mfma_products_per_fp32_product": products,
                        # per shape: launches per step, mean ms per launch, algorithmic TFLOP/s
                        "per_shape": {"_".join(str(x) for x in k): [round(per_step(k)[1], 2), round(per_step(k)[0] / max(per_step(k)[1], 1e-9), 4),
                                                                    round(gemm_flops(k) * per_step(k)[1] / (per_step(k)[0] * 1e-3) / 1e12, 1)]
                                      for k in sorted(gemm_keys, key=lambda kk: -per_step(kk)[0])},
                        "measured_in": where}

                # ---- message passing (every hand-written non-GEMM kernel) ---------------------------------------------
                mp_keys = [k for k in kt_all if k[0] not in GEMM_NAMES and k[0] not in HELPER_NAMES]
                mp_ms = sum(per_step(k)[0] for k in mp_keys)
                mp_bytes = sum(algorithmic_bytes(k) * per_step(k)[1] for k in mp_keys)
                gat_keys = [k for k in mp_keys if k[0].startswith(GAT_PREFIXES)]
                k123 = None
                if gat_keys:
                    k_ms = sum(per_step(k)[0] for k in gat_keys)
                    k_bytes = sum(algorithmic_bytes(k) * per_step(k)[1] for k in gat_keys)
                    # the survey's figure needs (H, D) per layer: the aggregate-first output layer (F -> H x D) is entered with
                    # its projected width, as the survey counts it
                    fwd_layers = []
                    for k in gat_keys:
                        base = k[0][:-5] if k[0].endswith("_bf16") else k[0]
                        if base == "gat_fwd":
                            fwd_layers.append(((k[0],) + tuple(k[1:5]), per_step(k)[1]))
                        elif base == "lspe_fwd":         # a fused level = a two-head structure layer + a one-head position layer
                            fwd_layers.append((("gat_fwd", k[1], k[2], 2, k[3]), per_step(k)[1]))
                            fwd_layers.append((("gat_fwd", k[1], k[2], 1, k[3]), per_step(k)[1]))
                        elif base == "gat_agg_fwd":
                            fwd_layers.append((("gat_fwd", k[1], k[2], k[3], cfg.MODEL.get("node_embed_dim", 1024)), per_step(k)[1]))
                    sv_bytes = survey_k123_bytes(fwd_layers, s_row)
                    k123 = {"kernels": sorted({k[0] for k in gat_keys}), "ms_per_step": k_ms,
                            "layer_edges_per_s": E * L / (k_ms * 1e-3),
                            "own_algorithmic_GB_per_step": k_bytes / 1e9, "own_achieved_GBps": k_bytes / (k_ms * 1e-3) / 1e9,
                            "own_frac_of_hbm_peak": k_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                            "survey_algorithmic_GB_per_step": sv_bytes / 1e9,
                            "survey_roofline_layer_edges_per_s": E * L / (sv_bytes / (HBM_PEAK_GBPS * 1e9)) if sv_bytes else None,
                            "frac_of_survey_roofline": (sv_bytes / (HBM_PEAK_GBPS * 1e9)) / (k_ms * 1e-3) if sv_bytes else None,
                            "traffic_GB_per_step": (lambda t: t / 1e9 if t is not None else None)(traffic_of(gat_keys)),
                            "measured_in": where}
                    if in_step and sv_bytes:
                        k123["in_step_ms_per_step"] = in_step["ms_per_step"]
                        k123["in_step_frac_of_survey_roofline"] = (sv_bytes / (HBM_PEAK_GBPS * 1e9)) / (in_step["ms_per_step"] * 1e-3)
                        k123["in_step_side_stream"] = in_step["side_stream"]
                    out["roofline_k123"] = k123
                out["message_passing"] = {"ms_per_step": mp_ms, "share_of_step": mp_ms / ms,
                                          "algorithmic_GB_per_step": mp_bytes / 1e9,
                                          "achieved_GBps": mp_bytes / (mp_ms * 1e-3) / 1e9 if mp_ms else None,
                                          "frac_of_hbm_peak": mp_bytes / (mp_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if mp_ms else None,
                                          "layer_edges_per_s_mp_only": E * L / (mp_ms * 1e-3) if mp_ms else None,
                                          "measured_in": f"{nprobe} fully instrumented warm-up step(s); GAT kernels: " + where,
                                          "per_kernel_ms": {"_".join(str(x) for x in k): round(per_step(k)[0] / max(per_step(k)[1], 1e-9), 5)
                                                            for k in sorted(mp_keys)}}

                # ---- the roofline object: fixed kernel per dtype ------------------------------------------------------
                rkeys = [k for k in kt_all if k[0] in roof_names]
                if rkeys:
                    r_ms = sum(per_step(k)[0] for k in rkeys)
                    r_n = sum(per_step(k)[1] for k in rkeys)
                    if bf16 or spmm_model:
                        r_bytes = sum(algorithmic_bytes(k) * per_step(k)[1] for k in rkeys)
                        ach = r_bytes / (r_ms * 1e-3) / 1e9
                        tr = traffic_of(rkeys)
                        what = "spgnn_gat_{fwd,bwd_dst,bwd_src}_bf16 + spgnn_gat_agg_{fwd,bwd_dst,bwd_src}_bf16: all launches of a step" if bf16 else \
                               " + ".join(sorted({"spgnn_" + k[0] for k in rkeys})) + ": all launches of a step (fwd + bwd)"
                        out["roofline"] = {"bound": "hbm", "kernel": what, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                           "frac": ach / HBM_PEAK_GBPS, "traffic": tr, "traffic_source": TRAFFIC_SOURCE if tr is not None else None,
                                           "frac_of_copy_bandwidth": ach / copy_bw["GBps"] if copy_bw else None,
                                           "algorithmic_bytes_per_step": r_bytes, "ms_per_step": r_ms, "launches_per_step": r_n,
                                           "avg_launch_ms": r_ms / r_n, "measured_in": where}
                    else:
                        r_fl = sum(gemm_flops(k) * per_step(k)[1] for k in rkeys)
                        ach = r_fl / (r_ms * 1e-3) / 1e12
                        out["roofline"] = {"bound": "mfma", "kernel": "spgnn_gemm_nt(_pair): all launches of a step (forward projections + input gradients)", "achieved": ach, "peak": MFMA_F16_PEAK, "unit": "TFLOP/s",
                                           "frac": ach / MFMA_F16_PEAK, "executed_mfma_frac": 3.0 * ach / MFMA_F16_PEAK,
                                           "frac_of_fp32_matrix_peak": ach / FP32_MATRIX_PEAK, "traffic": traffic_of(rkeys), "traffic_source": TRAFFIC_SOURCE,
                                           "algorithmic_flops_per_step": r_fl, "ms_per_step": r_ms, "launches_per_step": r_n,
                                           "avg_launch_ms": r_ms / r_n, "measured_in": where,
                                           "note": "achieved = ALGORITHMIC flops (2MNK of the fp32 product) / time; the kernel executes "
                                                   "three fp16 MFMA products per fp32 product (hi*hi + hi*lo + lo*hi): "
                                                   "executed_mfma_frac = 3 x frac; against the native fp32 matrix pipe (157.3 TFLOP/s) "
                                                   "the same rate is frac_of_fp32_matrix_peak"}
                        if k123 is not None:
                            # the north-star figure (BASELINE.json: message passing vs the HBM roofline) inside the object the
                            # driver keeps: SURVEY 8d's K1-K3 = every lspe_* / gat_* launch of a step
                            out["roofline"]["hbm"] = {
                                "bound": "hbm", "kernels": "K1-K3: " + " ".join(k123["kernels"]),
                                "ms_per_step": k123["ms_per_step"], "survey_bytes_per_step": sv_bytes, "own_bytes_per_step": k_bytes,
                                "achieved": sv_bytes / (k123["ms_per_step"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": k123["frac_of_survey_roofline"], "own_frac": k123["own_frac_of_hbm_peak"],
                                "layer_edges_per_s": k123["layer_edges_per_s"],
                                "traffic": traffic_of(gat_keys), "launches_per_step": sum(per_step(k)[1] for k in gat_keys)}
                            if in_step:
                                out["roofline"]["hbm"]["in_step_ms_per_step"] = in_step["ms_per_step"]
                                out["roofline"]["hbm"]["in_step_frac"] = (sv_bytes / (HBM_PEAK_GBPS * 1e9)) / (in_step["ms_per_step"] * 1e-3)
                                out["roofline"]["hbm"]["in_step_side_stream"] = in_step["side_stream"]

                # ---- composite roofline (SURVEY.md 8d / BASELINE.md 2): t_graph + max(t_gemm_bytes, t_gemm_flops) -----
                t_graph = mp_bytes / (HBM_PEAK_GBPS * 1e9) * 1e3
                t_gb = g_by / (HBM_PEAK_GBPS * 1e9) * 1e3
                t_gf32 = g_fl / (FP32_MATRIX_PEAK * 1e12) * 1e3
                t_gf16 = g_fl / (MFMA_F16_PEAK * 1e12) * 1e3
                comp_hbm = t_graph + t_gb
                comp_f32 = t_graph + max(t_gb, t_gf32)
                comp_16 = t_graph + max(t_gb, t_gf16 * products)
                out["composite"] = {"t_graph_ms": t_graph, "t_gemm_bytes_ms": t_gb, "t_gemm_flops_fp32_matrix_ms": t_gf32,
                                    "t_gemm_flops_16bit_mfma_ms": t_gf16 * products,
                                    "hbm_only_ms": comp_hbm, "fp32_matrix_ms": comp_f32, "as_executed_ms": comp_16,
                                    "step_vs_hbm_only": comp_hbm / ms, "step_vs_fp32_matrix": comp_f32 / ms, "step_vs_as_executed": comp_16 / ms,
                                    "layer_edges_per_s_at_hbm_only": E * L / (comp_hbm * 1e-3),
                                    "layer_edges_per_s_at_fp32_matrix": E * L / (comp_f32 * 1e-3),
                                    "note": "roofline step time = message-passing algorithmic bytes at 8 TB/s + max(GEMM algorithmic bytes at "
                                            "8 TB/s, GEMM flops at the named matrix peak); 'as_executed' prices the flops at the 2.5 PFLOP/s "
                                            f"16-bit MFMA peak times the {products} product(s) this precision executes per algorithmic product; "
                                            "step_vs_* = that time / the measured ms_per_step"}
            except Exception as e:
                out["accounting_error"] = repr(e)[:300]
        if capture_error:
            out["config"]["capture_error"] = capture_error
        flatten_for_driver(out)
        if eager_leg:
            eager_leg["value"] = E_all * L / (eager_leg["ms_per_step"] * 1e-3) if world == 1 else None
            eager_leg["note"] = "same step issued eagerly after the timed region (host-paced when the host is slower than the GPU)"
            out["eager"] = eager_leg
        return out, (cfg, model, samples)
    return None, (cfg, model, samples)


def flatten_for_driver(out):
    """Repeat the nested figures of a leg as flat scalars of the objects the driver's record keeps (`roofline`, `config`)."""
    r = out.get("roofline")
    if isinstance(r, dict):
        h = r.get("hbm")
        if isinstance(h, dict):          # f32 headline: the MFMA object carries the north-star K1-K3 figure
            r["hbm_frac"] = h["frac"]
            r["hbm_own_frac"] = h["own_frac"]
            r["hbm_ms_per_step"] = h["ms_per_step"]
            r["hbm_survey_bytes"] = h["survey_bytes_per_step"]
            r["hbm_own_bytes"] = h["own_bytes_per_step"]
            r["hbm_achieved_GBps"] = h["achieved"]
            r["hbm_traffic"] = h["traffic"]
            r["hbm_layer_edges_per_s"] = h["layer_edges_per_s"]
            r["hbm_launches_per_step"] = h["launches_per_step"]
            if "in_step_frac" in h:
                r["hbm_in_step_frac"] = h["in_step_frac"]
                r["hbm_in_step_ms_per_step"] = h["in_step_ms_per_step"]
    k = out.get("roofline_k123")
    if isinstance(k, dict) and isinstance(r, dict) and "hbm_frac" not in r:
        r["k123_frac_of_survey_roofline"] = k["frac_of_survey_roofline"]
        r["k123_ms_per_step"] = k["ms_per_step"]
        if "in_step_frac_of_survey_roofline" in k:
            r["k123_in_step_frac_of_survey_roofline"] = k["in_step_frac_of_survey_roofline"]
            r["k123_in_step_ms_per_step"] = k["in_step_ms_per_step"]
    c = out["config"]
    for name, key in (("gemm", "ms_per_step"), ("message_passing", "ms_per_step")):
        if isinstance(out.get(name), dict):
            c[f"{name}_ms_per_step"] = out[name][key]
    if isinstance(out.get("step_ms"), dict):
        c["step_ms_median"] = out["step_ms"]["median"]
    if isinstance(out.get("copy_bandwidth"), dict):
        c["copy_bandwidth_GBps"] = out["copy_bandwidth"]["GBps"]


LINE_LIMIT = 12288      # bytes: r04's 19.6 KB line was parsed by the driver, r05's 31.3 KB line was not; stay far below both


def _num(x, digits=6):
    """A scalar for the line: floats to ``digits`` significant digits, NaN / inf -> None (strict JSON), the rest as it is."""
    if isinstance(x, str):
        return x[:118]
    if isinstance(x, bool) or x is None or isinstance(x, int):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    return str(x)[:100]


def driver_line(out):
    """The ONE stdout line: the contract's top-level keys + `config`, `roofline`, `cpu_baseline` as FLAT SCALARS only (the
    driver's record keeps scalar members and cuts strings at 120 characters).  No prose members, no nested legs: the full
    object (every leg, per-shape tables, notes) goes to bench_detail.json.  Bounded by LINE_LIMIT (tests/test_host.py)."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
           "dtype", "data", "graph_edges_per_s", "loss", "dry_run", "accounting_error")
    line = {k: _num(out[k], 9) for k in top if k in out}
    c_in = out.get("config") or {}
    c = {k: _num(v) for k, v in c_in.items() if not isinstance(v, (dict, list, tuple))}
    if isinstance(c.get("workload"), str):
        c["workload"] = c["workload"][:118]
    if isinstance(c.get("gemm"), str):
        c["gemm"] = c["gemm"][:80]
    sm = out.get("step_ms")
    if isinstance(sm, dict):
        c["step_ms_median"], c["step_ms_p10"], c["step_ms_p90"] = _num(sm.get("median")), _num(sm.get("p10")), _num(sm.get("p90"))
    comp = out.get("composite")
    if isinstance(comp, dict):
        for k in ("hbm_only_ms", "as_executed_ms", "fp32_matrix_ms", "step_vs_hbm_only", "step_vs_as_executed"):
            c["composite_" + k] = _num(comp.get(k))
    mp = out.get("message_passing")
    if isinstance(mp, dict):
        c["message_passing_frac_of_hbm_peak"] = _num(mp.get("frac_of_hbm_peak"))
    if isinstance(out.get("eager"), dict):
        c["eager_ms_per_step"] = _num(out["eager"].get("ms_per_step"))
    line["config"] = c
    r_in = out.get("roofline")
    if isinstance(r_in, dict):
        skip = ("traffic_source", "measured_in", "note")
        r = {k: _num(v) for k, v in r_in.items() if not isinstance(v, (dict, list, tuple)) and k not in skip}
        if isinstance(r.get("kernel"), str):
            r["kernel"] = r["kernel"][:118]
        r["traffic_measured_in_this_run"] = False          # profiles/traffic_latest.json: the builder's rocprofv3 --pmc passes
        line["roofline"] = r
    b_in = out.get("cpu_baseline")
    if isinstance(b_in, dict):
        b = {k: _num(v) for k, v in b_in.items() if not isinstance(v, (dict, list, tuple)) and k != "sample64"}
        if isinstance(b.get("sample"), str):
            b["sample"] = b["sample"][:118]
        line["cpu_baseline"] = b
    line["detail"] = out.get("detail_file", "bench_detail.json")
    txt = json.dumps(line, allow_nan=False, separators=(", ", ": "))
    if len(txt) > LINE_LIMIT:                               # never outgrow the driver again: shed the optional scalars, longest first
        for part in ("config", "roofline"):
            for k in sorted(line.get(part, {}), key=lambda q: -len(q)):
                if len(txt) <= LINE_LIMIT:
                    break
                if k.startswith(("sec_", "composite_", "batch_cycle_", "hbm_own", "comm_library")):
                    del line[part][k]
                    txt = json.dumps(line, allow_nan=False, separators=(", ", ": "))
    return txt


def write_detail(out, path):
    """The full object of the run (every leg, nested) as a file beside the line; -> the path written, or None."""
    for cand in (path, os.path.join(ROOT, "bench_detail.json"), os.path.join(os.environ.get("TMPDIR", "/tmp"), "bench_detail.json")):
        if not cand:
            continue
        try:
            with open(cand, "w") as f:
                json.dump(out, f)
                f.write("\n")
            return cand
        except OSError:
            continue
    return None


def batch_cycle(dev, config="st_pgat_spgnn_3", trees=64, n_batches=6, inner=300, granule=512):
    """The reference's loader-batch cycle as a measured quantity (job_runner.py:1870-1920, exp_settings/st_pgat_spgnn_3.py:29,34:
    GCN_STEPS = 300 optimizer steps on every freshly built batch of TRAIN_BATCH_SIZE = 64 trees): per batch assemble
    (data.assemble_batch: pinned packing, upload, device CSC, device anchors + distance encoding), load into the batch arena of
    its size class (spgnn_amd/arena.py; the first batch of a class also pays warm-up + capture) and ``inner`` steps as HIP-graph
    replays.  Reported beside it: the r3 flow (a fresh capture for every batch) and the steady-state replay of an unpadded
    64-tree batch, which the amortised figure is a multiple of."""
    from spgnn_amd import data, models, synthetic
    from spgnn_amd.configs import class_weight_list, get_config
    from spgnn_amd.train import TrainStep
    cfg = get_config(config)
    torch.manual_seed(0)

    def fresh_step():
        model = models.build_model(cfg.MODEL).to(dev)
        model.init(None); model.set_gcn_only(); model.train()
        return TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, cfg.OPTIMIZER["lr"], cfg.OPTIMIZER["momentum"], seed=99)

    def wall(fn):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(dev)
        return r, (time.perf_counter() - t0) * 1e3

    batches = [synthetic.synthetic_trees(trees, rank=200 + b) for b in range(n_batches)]     # the dataset: host trees, not timed
    # steady state: an unpadded batch, captured, replayed
    ts0 = fresh_step()
    g0, _ = wall(lambda: data.assemble_batch(batches[0], dev, cfg.POS_ENC_DIM))
    ts0.capture(g0)
    for _ in range(10):
        ts0.replay()
    _, t = wall(lambda: [ts0.replay() for _ in range(inner)])
    steady = t / inner
    del ts0, g0
    # the arena flow
    ts = fresh_step()
    per_batch = []
    for b, samples in enumerate(batches):
        g, asm = wall(lambda: data.assemble_batch(samples, dev, cfg.POS_ENC_DIM))
        known = len(ts._captures)
        ag, load = wall(lambda: ts.arena_graph(g, granule))
        cap = 0.0
        done = 0
        if not ts.select(ag):
            _, cap = wall(lambda: ts.capture(ag))
            done = ts.capture_steps
        _, rep = wall(lambda: [ts.replay() for _ in range(inner - done)])
        per_batch.append({"nodes": g.number_of_nodes(), "class_nodes": ag.number_of_nodes(), "new_class": len(ts._captures) > known,
                          "assemble_ms": asm, "arena_load_ms": load, "capture_ms": cap, "replay_ms_per_step": rep / (inner - done),
                          "amortised_ms_per_step": (load + cap + rep) / inner,
                          "amortised_incl_assemble_ms_per_step": (asm + load + cap + rep) / inner})
        del g
    # the whole loop as a training job runs it (TrainStep.run_batches): the assembly of batch i + 1 on a side stream under
    # the replays of batch i.  Same batches again - their classes are known now - one wall clock around everything.
    _, pipe = wall(lambda: ts.run_batches(batches, inner, lambda s_: data.assemble_batch(s_, dev, cfg.POS_ENC_DIM), granule))
    pipelined = pipe / (inner * len(batches))
    hits = [q for q in per_batch if not q["new_class"]]
    miss = [q for q in per_batch if q["new_class"]]
    mean = lambda xs: sum(xs) / len(xs) if xs else None
    # the r3 flow on the same batches: warm-up + capture + instantiate for every batch
    ts2 = fresh_step()
    rec = []
    for samples in batches[:3]:
        g = data.assemble_batch(samples, dev, cfg.POS_ENC_DIM)
        _, cap = wall(lambda: ts2.capture(g))
        _, rep = wall(lambda: [ts2.replay() for _ in range(inner - ts2.capture_steps)])
        rec.append({"capture_ms": cap, "amortised_ms_per_step": (cap + rep) / inner})
        del g
    return {"workload": f"{config}, {n_batches} loader batches of {trees} synthetic trees, {inner} optimizer steps on each (reference GCN_STEPS), "
                        f"fp32, dropout on, arena granule {granule} nodes",
            "steady_state_ms_per_step": steady, "classes_captured": len(ts._captures), "batches": per_batch,
            "amortised_ms_per_step_known_class": mean([q["amortised_ms_per_step"] for q in hits]),
            "amortised_ms_per_step_new_class": mean([q["amortised_ms_per_step"] for q in miss]),
            "amortised_over_steady_known_class": (mean([q["amortised_ms_per_step"] for q in hits]) / steady) if hits else None,
            "amortised_incl_assemble_ms_per_step_known_class": mean([q["amortised_incl_assemble_ms_per_step"] for q in hits]),
            "pipelined_loop": {"what": "TrainStep.run_batches over the same batches (classes known): assembly of batch i+1 under the "
                                       "replays of batch i; wall time of the whole loop / steps, assembly INCLUDED",
                               "ms_per_step": pipelined, "over_steady": pipelined / steady},
            "assemble_ms": mean([q["assemble_ms"] for q in per_batch]), "arena_load_ms": mean([q["arena_load_ms"] for q in per_batch]),
            "capture_ms": mean([q["capture_ms"] for q in miss]),
            "recapture_every_batch": {"capture_ms": mean([q["capture_ms"] for q in rec]),
                                      "amortised_ms_per_step": mean([q["amortised_ms_per_step"] for q in rec]),
                                      "amortised_over_steady": mean([q["amortised_ms_per_step"] for q in rec]) / steady}}


def single_tree_forward(dev, config="st_pgat_spgnn_3", sizes=(150, 300), reps=200, count_launches=False):
    """The reference's per-scan inference pattern as a measured quantity (job_runner.py:2046-2052, 1601-1610: one graph per
    scan, dgl.batch([g]), ONE model.forward(g); README.md:49-51 quotes per-scan test times): a single synthetic airway tree
    of n branches, eval mode, forward only, fp32.  Per size: the eagerly issued forward (host-paced: ~40 launches) and the
    captured forward of spgnn_amd.infer.ForwardRunner (one HIP-graph replay per scan of a size class, arena load included),
    median latency over ``reps`` scans; the launches of one forward - how many are this library's kernels (ops.KernelTimer
    sees every launch through the C ABI) out of all device kernels (torch.profiler) - and the CPU oracle's time for the same
    scan beside it."""
    import statistics
    from oracle import dgl_cpu as O
    from spgnn_amd import models, ops, synthetic
    from spgnn_amd.configs import get_config
    from spgnn_amd.infer import ForwardRunner
    cfg = get_config(config)
    torch.manual_seed(0)
    model = models.build_model(cfg.MODEL).to(dev)
    model.init(None); model.set_gcn_only(); model.eval()
    runner = ForwardRunner(model, granule=64)
    out = {"workload": f"{config}, ONE synthetic tree per forward (reference per-scan inference), eval mode, fp32, forward only",
           "sizes": {}}
    for n in sizes:
        scans = [synthetic.make_batch(1, rank=700 + i, device=dev, pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None), fixed_n=n) for i in range(4)]
        for g in scans:
            g.csc(dev)
        with torch.no_grad():
            for g in scans:
                model(g); runner(g)
            torch.cuda.synchronize(dev)

            def lat(fn):
                ts = []
                for i in range(reps):
                    g = scans[i % len(scans)]
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    fn(g)
                    torch.cuda.synchronize(dev)
                    ts.append((time.perf_counter() - t0) * 1e6)
                return statistics.median(ts), pct(ts, 0.9)
            eager_us, eager_p90 = lat(model)
            cap_us, cap_p90 = lat(runner)
            # device time of the replay alone (no arena load, no host): HIP events around back-to-back replays
            arena, graph = next(reversed(runner._classes.values()))[:2]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                graph.replay()
            e1.record(); torch.cuda.synchronize(dev)
            replay_us = e0.elapsed_time(e1) / 50 * 1e3
            # launches of ONE forward: the library's own (through the C ABI) and all device kernels
            ops.KernelTimer.start()
            model(scans[0])
            ops.KernelTimer.stop()
            lib_launches = len(ops.KernelTimer.sequence)
            total_launches = None
            try:
                if not count_launches:          # the default run does not pay for a torch.profiler pass (tools/infer_trace.py counts)
                    raise RuntimeError("not counted")
                from torch.profiler import ProfilerActivity, profile
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    model(scans[0])
                    torch.cuda.synchronize(dev)
                total_launches = sum(1 for ev in prof.events() if str(getattr(ev, "device_type", "")).endswith("CUDA"))
            except Exception:
                total_launches = None
            # parity and the CPU oracle's time on the same scan
            g = scans[0]
            outs = runner(g)
            src, dst = g.cpu().edges()
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            pe = g.ndata["pos_enc"].cpu() if "pos_enc" in g.ndata else None
            ref = O.net_forward(cfg.KIND, sd, src, dst, g.number_of_nodes(), g.ndata["fvs"].cpu(), pe)[0]
            err = float((outs[0].cpu() - ref).abs().max() / ref.abs().max())
            cts = []
            for _ in range(5):
                t0 = time.perf_counter()
                O.net_forward(cfg.KIND, sd, src, dst, g.number_of_nodes(), g.ndata["fvs"].cpu(), pe)
                cts.append((time.perf_counter() - t0) * 1e3)
        out["sizes"][str(n)] = {"nodes": n, "edges": g.number_of_edges(), "eager_us": round(eager_us, 1), "eager_us_p90": round(eager_p90, 1),
                                "captured_us": round(cap_us, 1), "captured_us_p90": round(cap_p90, 1), "replay_device_us": round(replay_us, 1),
                                "library_launches": lib_launches, "device_kernel_launches": total_launches,
                                "library_launch_fraction": (round(lib_launches / total_launches, 3) if total_launches else None),
                                "logits_rel_err_vs_oracle": err, "cpu_oracle_ms": round(statistics.median(cts), 2)}
    first = out["sizes"][str(sizes[0])]
    out.update({"replay_device_us": first["replay_device_us"], "captured_us": first["captured_us"], "eager_us": first["eager_us"],
                "cpu_oracle_ms": first["cpu_oracle_ms"],
                "what": "replay_device_us: the captured forward itself (HIP events around back-to-back graph replays); captured_us: one scan "
                        "through ForwardRunner on the host clock, arena load (two launches: spgnn_arena_load + spgnn_ell_rows_both) included; eager_us: model(g) issued eagerly"})
    return out


# BASELINE.json configs 2-4 beside the headline (config 5 at N = 1): run after it, short, inside the same JSON line
SECONDARY_LEGS = (("st_gat_6_bf16_512", "st_gat_6", "bf16", 512),
                  ("st_pgat_spgnn_3_f32_64", "st_pgat_spgnn_3", "f32", 64),
                  ("st_gat_3_f32_64", "st_gat_3", "f32", 64))
SECONDARY_KEYS = ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "step_ms", "roofline", "roofline_k123", "loss")


def secondary_summary(out):
    """What a secondary leg contributes to the line: its time, its launch mode and its roofline objects."""
    s = {k: out[k] for k in SECONDARY_KEYS if k in out}
    s["launch"] = out["config"]["launch"]
    s["workload"] = out["config"]["workload"]
    s["nodes"], s["edges"], s["conv_layers"] = out["config"]["nodes"], out["config"]["edges"], out["config"]["conv_layers"]
    for k in ("gemm", "message_passing"):
        if k in out:
            s[k] = {q: out[k][q] for q in ("ms_per_step", "frac_of_hbm_peak", "achieved_GBps", "algorithmic_TFLOPs",
                                          "executed_mfma_TFLOPs") if q in out[k]}
    return s


def visible_gpus():
    """GPUs this process may use, counted WITHOUT any torch.cuda / HIP call (the launcher must never bring up the runtime:
    torch.cuda.device_count() falls back to hipGetDeviceCount when amdsmi does not answer): the KFD topology lists every
    agent, GPUs are the nodes with SIMDs; a *_VISIBLE_DEVICES list narrows it.  None: cannot tell (the ranks will)."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for path in nodes:
        try:
            for ln in open(path):
                t = ln.split()
                if len(t) == 2 and t[0] == "simd_count" and int(t[1]) > 0:
                    n += 1
        except OSError:
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n: int) -> int:
    """`python3 bench.py --gpus N` without torch.distributed.run: start the N ranks ourselves - fresh child processes of this
    same command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly the environment torch.distributed.run
    would give them - BEFORE this process has made any GPU call (it never does: a process that has initialised the GPU must
    not exec, and need not here).  Rank 0's stdout is passed through (the ONE JSON line); the other ranks' stdout goes to
    stderr.  Returns the first non-zero exit code of a rank (the rest are then stopped by PID), else 0."""
    import signal
    import socket
    import subprocess
    import threading
    rehearsal = os.environ.get("SPGNN_BENCH_REHEARSAL", "0") == "1" or os.environ.get("SPGNN_BENCH_DRY", "0") == "1"
    if not rehearsal:
        have = visible_gpus()                   # /sys only: the launcher never brings up the GPU runtime
        if have is not None and have < n:
            print(f"bench.py: --gpus {n} but {have} GPU(s) visible", file=sys.stderr)
            return 2
    with socket.socket() as sk:                 # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPGNN_BENCH_CHILD="1")
        env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores(256) // n)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))

    def pump(pipe):                              # rank 0's lines as they come: JSON to stdout, library chatter (gloo prints its
        for line in pipe:                        # connection notes on stdout) to stderr, so stdout stays the ONE line
            dst = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            dst.write(line)
            dst.flush()
    t = threading.Thread(target=pump, args=(procs[0].stdout,), daemon=True)
    t.start()

    def stop_all(sig=signal.SIGTERM):
        for q in procs:
            if q.poll() is None:
                try:
                    q.send_signal(sig)           # exact PIDs of the children started above
                except ProcessLookupError:
                    pass
    signal.signal(signal.SIGTERM, lambda *_: (stop_all(), sys.exit(143)))
    rc, deadline = 0, None
    try:
        while any(q.poll() is None for q in procs):
            for q in procs:
                c = q.poll()
                if c not in (None, 0) and rc == 0:
                    rc, deadline = c, time.time() + 15.0      # a rank died: its peers would wait in a collective forever
            if deadline is not None and time.time() > deadline:
                stop_all()
                time.sleep(3.0)
                stop_all(signal.SIGKILL)
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        stop_all()
        rc = 130
    for q in procs:
        c = q.wait()
        if c != 0 and rc == 0:
            rc = c
    t.join(timeout=5.0)
    return rc


def dry_rank(args, rank, world):
    """SPGNN_BENCH_DRY=1: the launch / rendezvous / one-line protocol of an N-rank run WITHOUT a GPU (gloo on CPU tensors), so
    the self-launch path has a CPU test (tests/test_host.py).  The line says it is a dry run and carries no measurement."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    ones = torch.ones(1)
    dist.all_reduce(ones)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "message-passing edges/sec (fwd+bwd), batched trees", "value": None, "unit": "layer-edges/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "config": {"comm_world_size": int(ones.item()), "comm_backend": "gloo (dry run, no GPU work)",
                                     "launched_by": "bench.py self_launch" if os.environ.get("SPGNN_BENCH_CHILD") else "torch.distributed.run"}}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="st_pgat_spgnn_3")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="storage dtype of node-feature rows inside the GNN head")
    ap.add_argument("--trees", type=int, default=512, help="trees per GPU")
    ap.add_argument("--loss-rows-only", action="store_true",
                    help="TrainStep(loss_rows_only=True): output-layer projection, classifier and their backward products on the rows the "
                         "step's mask keeps only (same loss and gradients; NOT the headline: the headline runs every row)")
    ap.add_argument("--loss-rows-backward", action="store_true",
                    help='TrainStep(loss_rows_only="backward"): dense forward, only the BACKWARD products of that part on the kept rows '
                         "(the rows skipped are exactly zero in the dense step)")
    ap.add_argument("--heads", type=int, default=0, help="override the hidden GAT layers' head count (0: the config's own)")
    ap.add_argument("--eager", action="store_true",
                    help="time eagerly issued steps instead of HIP-graph replays of the step (TrainStep.capture)")
    ap.add_argument("--no-eager-leg", action="store_true", help="graph mode: skip the eager steps after the timed region "
                    "(they carry the HIP events around the roofline kernels)")
    ap.add_argument("--no-dropout", action="store_true", help="eval-mode arithmetic (parity runs); default keeps dropout on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short legs for BASELINE configs 2-4 after the headline")
    ap.add_argument("--cpu-trees", type=int, default=0, help="trees of the cpu_baseline sample (0: the whole batch, i.e. the headline workload)")
    ap.add_argument("--batch-cycle-only", action="store_true", help="run only the loader-batch cycle leg (secondary.batch_cycle_64) and print it")
    ap.add_argument("--single-tree-only", action="store_true", help="run only the per-scan inference leg (secondary.single_tree_forward) and print it")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-side-stream", action="store_true", help="ops.OVERLAP_TN off: every launch of the timed step alone on its stream "
                    "(how roofline.hbm_frac is measured; the profile of such a run makes that figure reproducible from rocprofv3 --stats)")
    ap.add_argument("--full-line", action="store_true", help="print the FULL nested object on stdout (tools/; tens of KB) instead of the "
                    "driver's flat line; the default writes it to --detail-out only")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "bench_detail.json"), help="where the full nested object of the run goes")
    ap.add_argument("--cpu-small-sample", action="store_true", help="cpu_baseline: also time the 64-tree sample of rounds 1-4 (value64)")
    ap.add_argument("--count-launches", action="store_true", help="single-tree leg: count all device launches of a forward with torch.profiler")
    ap.add_argument("--graph", action="store_true", help="(default) kept for older command lines")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SPGNN_BENCH_FORCE_LAUNCH=1: take the launcher path at ANY N, and keep the rank-exchange code on with one rank (RCCL at
    # world size 1, the two-graph step, the real TrainStep._reduce) - how a one-GPU box exercises the N > 1 code path
    forced = os.environ.get("SPGNN_BENCH_FORCE_LAUNCH", "0") == "1"
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or forced):
        # plain `python3 bench.py --gpus N`: this process becomes the launcher of N ranks and never touches the GPU itself
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        args.gpus = world                       # under torch.distributed.run the launcher's world size is the truth
    if os.environ.get("SPGNN_BENCH_DRY", "0") == "1":
        return dry_rank(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the message-passing path has no CPU fallback)")
    # Rehearsal of the N > 1 flow on a one-GPU box: SPGNN_BENCH_REHEARSAL=1 puts every rank on device 0 and moves the
    # tensors with gloo (RCCL refuses two ranks per device).  Never set by the driver; numbers from it mean nothing.
    rehearsal = os.environ.get("SPGNN_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    exchange = world > 1 or (forced and "WORLD_SIZE" in os.environ)
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    if args.no_side_stream:
        from spgnn_amd import ops as _ops
        _ops.OVERLAP_TN = False
    if args.batch_cycle_only:
        print(json.dumps({"batch_cycle_64": batch_cycle(dev)}), flush=True)
        return
    if args.single_tree_only:
        print(json.dumps({"single_tree_forward": {c_: single_tree_forward(dev, c_, count_launches=args.count_launches) for c_ in ("st_pgat_spgnn_3", "st_gat_3", "st_gcn_3", "st_gin_3", "st_sage_3")}}), flush=True)
        return
    out, (cfg, model, samples) = run_leg(args.config, args.dtype, args.trees, args.steps, args.warmup, rank=rank, world=world, dev=dev,
                                         eager=args.eager, no_eager_leg=args.no_eager_leg, no_dropout=args.no_dropout,
                                         no_kernel_timers=args.no_kernel_timers, heads=args.heads, loss_rows=("backward" if args.loss_rows_backward else args.loss_rows_only),
                                         exchange=exchange)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, model, samples, min(args.cpu_trees, args.trees) if args.cpu_trees else args.trees,
                                               small_sample=args.cpu_small_sample)
        headline = args.config == "st_pgat_spgnn_3" and args.dtype == "f32" and args.trees == 512
        if world == 1 and headline and not args.no_secondary:
            del model, samples
            sec = {}
            for name, c_, d_, t_ in SECONDARY_LEGS:
                torch.cuda.empty_cache()
                try:
                    o2, _ctx = run_leg(c_, d_, t_, min(args.steps, 20), min(args.warmup, 5), rank=0, world=1, dev=dev,
                                       no_dropout=args.no_dropout, no_kernel_timers=args.no_kernel_timers,
                                       copy_bw=out.get("copy_bandwidth"))
                    sec[name] = secondary_summary(o2)
                    del o2, _ctx
                except Exception as e:                  # never lose the headline to a secondary leg
                    sec[name] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
            # the headline workload with TrainStep(loss_rows_only=...) - NOT the headline number: the headline runs every row of
            # every layer, forward and backward.  "backward": dense forward, the output layer's backward products on the rows the
            # step's mask keeps (the rows skipped are exactly zero in the dense step); True: the forward behind the last
            # aggregation on those rows too (no other row reaches the loss or a gradient).  Same loss, same gradients.
            for leg_name, mode in (("st_pgat_spgnn_3_f32_512_loss_rows_backward", "backward"), ("st_pgat_spgnn_3_f32_512_loss_rows_only", True)):
                try:
                    o2, _ctx = run_leg("st_pgat_spgnn_3", "f32", 512, min(args.steps, 20), min(args.warmup, 5), rank=0, world=1, dev=dev,
                                       no_dropout=args.no_dropout, no_kernel_timers=args.no_kernel_timers,
                                       copy_bw=out.get("copy_bandwidth"), loss_rows=mode)
                    s2 = secondary_summary(o2)
                    s2["what"] = ("the headline workload with TrainStep(loss_rows_only=%r): %s only on the rows the step's mask keeps "
                                  "(reference job_runner.py:1896-1900 takes the loss over pre[mask]); identical loss and gradients up to fp32 "
                                  "summation order (tests/test_hip_loss_rows.py); every traversal still visits every edge (the output layer's two backward "
                                  "traversals read the listed gradient rows only: their share of roofline_k123 is priced on the dense byte count)"
                                  % (mode, "the output layer's BACKWARD products (dense forward)" if mode == "backward" else
                                     "output-layer projection, head mean, classifier and their backward products"))
                    sec[leg_name] = s2
                    del o2, _ctx
                except Exception as e:
                    sec[leg_name] = {"error": repr(e)[:300]}
                torch.cuda.empty_cache()
            torch.cuda.empty_cache()
            try:
                sec["batch_cycle_64"] = batch_cycle(dev)
            except Exception as e:
                sec["batch_cycle_64"] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
            try:
                sec["single_tree_forward"] = single_tree_forward(dev, count_launches=args.count_launches)
            except Exception as e:
                sec["single_tree_forward"] = {"error": repr(e)[:300]}
            out["secondary"] = sec
            bc = sec["batch_cycle_64"]
            c = out["config"]
            if "error" not in bc:                       # flat: the driver's record keeps scalar members of `config` only
                c["batch_cycle_steady_ms"] = round(bc["steady_state_ms_per_step"], 4)
                c["batch_cycle_known_class_ms"] = round(bc["amortised_ms_per_step_known_class"] or 0.0, 4)
                c["batch_cycle_over_steady"] = round(bc["amortised_over_steady_known_class"] or 0.0, 4)
                c["batch_cycle_loop_incl_assembly_over_steady"] = round(bc["pipelined_loop"]["over_steady"], 4)
                c["batch_cycle_capture_ms"] = round(bc["capture_ms"] or 0.0, 2)
            for name, leg in sec.items():               # the secondary legs (BASELINE configs 2-4, single-scan inference) likewise
                if not isinstance(leg, dict) or "error" in leg:
                    continue
                if "ms_per_step" in leg:
                    c[f"sec_{name}_ms"] = round(leg["ms_per_step"], 4)
                rr = leg.get("roofline") or {}
                if "frac" in rr:
                    c[f"sec_{name}_roofline_{rr.get('bound', '')}_frac"] = round(rr["frac"], 4)
                if "hbm_frac" in rr:
                    c[f"sec_{name}_k123_hbm_frac"] = round(rr["hbm_frac"], 4)
                if "k123_frac_of_survey_roofline" in rr:
                    c[f"sec_{name}_k123_hbm_frac"] = round(rr["k123_frac_of_survey_roofline"], 4)
                if "executed_mfma_frac" in rr:
                    c[f"sec_{name}_executed_mfma_frac"] = round(rr["executed_mfma_frac"], 4)
                for q in ("replay_device_us", "captured_us", "eager_us", "cpu_oracle_ms"):
                    if q in leg:
                        c[f"sec_{name}_{q}"] = leg[q]
        detail = write_detail(out, args.detail_out)
        out["detail_file"] = os.path.basename(detail) if detail else "not written"
        if args.full_line:
            print(json.dumps(out), flush=True)
        else:
            print(driver_line(out), flush=True)
        print(f"bench.py: full object of this run -> {detail}", file=sys.stderr, flush=True)
    if exchange:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
