#!/usr/bin/env python3
"""bench.py — message-passing edges/sec (fwd+bwd) of the GNN training step on N MI355X.

One "step" = one optimizer step on one static batched graph of synthetic random-fan-out trees
(SURVEY.md Appendix D): Bernoulli node mask -> model forward -> class-weighted CE -> backward ->
[RCCL all-reduce of the flat gradient bucket] -> fused SGD(momentum).  Nothing is skipped inside
the timed region.  Default workload: st_pgat_spgnn_3 (full SPGNN + LSPE position stream), 512 trees
per GPU, fp32, dropout on — the configuration BASELINE.json's north_star quotes its target on; with
N GPUs every rank gets its own 512 trees (weak scaling, global batch 512*N).

Single GPU:  python bench.py [--steps K --warmup W]
Multi GPU:   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                 --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
"roofline" for the dominant hand-written kernel (algorithmic bytes per launch / mean launch time
from HIP events recorded on the launch stream during the timed region) and "cpu_baseline" (the
DGL-CPU-equivalent restatement under oracle/, timed on this host's cores on a bounded sample).
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

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def algorithmic_bytes(key) -> float:
    """Algorithmic HBM bytes of ONE launch (SURVEY.md §8d per-unit figures, fp32):
    every operand row read once, every result row written once, index arrays once."""
    name = key[0]
    if name == "gat_fwd":            # read ft [+res]; write out [and/or its head mean]; read el, er; write a; CSC
        _, N, E, H, D, has_res, mean, has_out = key
        return (4 * (1 + has_res + has_out) * N * H * D + 4 * mean * N * D + 4 * (2 * N * H + E * H)
                + 4 * (N + 1 + E))
    if name == "gat_bwd_dst":        # read g_out [,out], ft; write g_pre; read el, er, a; write g_e, g_er; CSC
        _, N, E, H, D, act, mean = key
        return (4 * (2 + (1 if act else 0)) * N * H * D + 4 * N * (D if mean else H * D)
                + 4 * (3 * N * H + 2 * E * H) + 4 * (N + 1 + E))
    if name == "gat_bwd_src":        # read g_pre, write g_ft; read a, g_e; write g_el; CSR + slot map
        _, N, E, H, D = key
        return 4 * 2 * N * H * D + 4 * (N * H + 2 * E * H) + 4 * (N + 1 + 2 * E)
    if name == "scores_fwd":         # read x; write S
        _, N, K, J = key
        return 4 * N * K + 4 * N * J
    if name == "scores_bwd_w":       # read x, gS; partials negligible
        _, N, K, J = key
        return 4 * N * K + 4 * N * J
    if name == "scores_bwd_x":       # read + write gX; read gS
        _, N, K, J = key
        return 4 * 2 * N * K + 4 * N * J
    if name == "gat_agg_fwd":        # read x; write the z blocks [with a copy of x per head]; el, er, a; CSC
        _, N, E, H, F_, xcopy = key
        return 4 * N * F_ + 4 * N * H * F_ * (2 if xcopy else 1) + 4 * (2 * N * H + E * H) + 4 * (N + 1 + E)
    if name == "gat_agg_bwd_dst":    # read the z part of g_z and x; el, er, a; write g_e, g_er; CSC
        _, N, E, H, F_ = key
        return 4 * N * (H + 1) * F_ + 4 * (3 * N * H + 2 * E * H) + 4 * (N + 1 + E)
    if name == "gat_agg_bwd_src":    # read g_z (both parts); write g_x; a, g_e, g_er, g_el; CSR + slot map
        _, N, E, H, F_ = key
        return 4 * N * (2 * H + 1) * F_ + 4 * (2 * N * H + 2 * E * H) + 4 * (N + 1 + 2 * E)
    if name == "head_mean":
        _, N, H, D = key
        return 4 * N * (H + 1) * D
    if name == "act_bwd":            # read g_out [and out]; write g_pre
        _, N, H, D, act, mean = key
        return 4 * N * (D if mean else H * D) + 4 * N * H * D * (2 if act else 1)
    if name == "act_bwd_proj":       # read g_logits and out; write g_pre (the classifier's input gradient stays in registers)
        _, N, H, D, act, J = key
        return 4 * N * H * D * (2 if act else 1) + 4 * N * J + 4 * J * D
    if name == "sum_partials":
        return 0.0
    if name == "scores_from_parts":      # read the (N, H*D/64, 2) partials, write (N, 2H)
        _, N, H, D = key
        return 4 * 2 * N * H * (D // 64) + 4 * 2 * N * H
    if name == "masked_ce":
        _, N, C = key
        return 4 * 2 * N * C + 4 * 4 * N
    if name in ("gemm_nt", "gemm_tn", "absmax", "split_rows"):
        return 0.0                    # compute-bound / helper kernels: reported in "gemm", not in the HBM accounting
    if name == "spmm_sum":
        _, N, E, F = key
        return 4 * 2 * N * F + 4 * (N + 1 + E)
    if name == "spmm_max_fwd":
        _, N, E, F = key
        return 4 * 3 * N * F + 4 * (N + 1 + E)
    if name == "spmm_max_bwd":
        _, N, E, F = key
        return 4 * 3 * N * F + 4 * (N + 1 + 2 * E)
    raise KeyError(name)


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


def cpu_baseline(cfg, model, samples, n_trees, steps=5, warm=1, budget_s=25.0):
    """DGL-CPU-equivalent (restated) fwd+bwd on the host cores: oracle/dgl_cpu.py, same weights, first
    ``n_trees`` trees of rank 0's batch, eval-mode arithmetic (no dropout), all host threads."""
    from oracle import dgl_cpu as O
    from spgnn_amd import synthetic
    from spgnn_amd.configs import class_weight_list
    import torch.nn.functional as F
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
    for i in range(warm + steps):
        if i > warm and time.perf_counter() - t_start > budget_s:
            break
        t0 = time.perf_counter()
        out = O.net_forward(cfg.KIND, sd, src, dst, n, g.ndata["fvs"], g.ndata.get("pos_enc"))[0]
        loss = O.masked_weighted_ce(out, y, mask, w)
        grads = torch.autograd.grad(loss, [p for p in sd.values() if p.requires_grad], allow_unused=True)
        del grads
        if i >= warm:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": E * cfg.CONV_LAYERS / med, "unit": "layer-edges/s", "cores": cores, "kind": "port",
            "sample": f"first {n_trees} trees of rank 0's batch (N={n}, E={E}), fwd+bwd, median of {len(times)} "
                      f"after {warm} warm-up(s) on {cores} threads, {med * 1e3:.1f} ms/iter, DGL-CPU-equivalent (restated) on torch CPU ops"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--config", default="st_pgat_spgnn_3")
    ap.add_argument("--trees", type=int, default=512, help="trees per GPU")
    ap.add_argument("--eager", action="store_true",
                    help="time eagerly issued steps instead of HIP-graph replays of the step (TrainStep.capture)")
    ap.add_argument("--no-eager-leg", action="store_true", help="graph mode: skip the eager steps after the timed region "
                    "(they carry the HIP events around the dominant kernel for the roofline object)")
    ap.add_argument("--no-dropout", action="store_true", help="eval-mode arithmetic (parity runs); default keeps dropout on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-trees", type=int, default=32)
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--graph", action="store_true", help="(default) kept for older command lines")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the message-passing path has no CPU fallback)")
    # Rehearsal of the N > 1 flow on a one-GPU box: SPGNN_BENCH_REHEARSAL=1 puts every rank on device 0 and moves the
    # tensors with gloo (RCCL refuses two ranks per device).  Never set by the driver; numbers from it mean nothing.
    rehearsal = os.environ.get("SPGNN_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    from spgnn_amd import _capi, models, ops, synthetic
    from spgnn_amd.configs import class_weight_list, get_config
    from spgnn_amd.train import TrainStep
    _capi.load()                                              # fail loudly if the HIP library is missing

    cfg = get_config(args.config)
    torch.manual_seed(0)                                      # identical replicas on every rank
    model = models.build_model(cfg.MODEL).to(dev)
    model.init(None)
    model.set_gcn_only()
    model.train(not args.no_dropout)
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)

    samples = synthetic.synthetic_trees(args.trees, rank=rank)
    g = synthetic.batch_from_samples(samples, dev, cfg.POS_ENC_DIM)
    g.csc(dev)                                                # CSC/CSR built once per loader batch (static for all steps)
    N, E = g.number_of_nodes(), g.number_of_edges()
    step = TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, cfg.OPTIMIZER["lr"],
                     cfg.OPTIMIZER["momentum"], seed=1234 + rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up.  Its last two steps are fully instrumented (HIP events around every hand-written kernel launch):
    # they give the per-kernel breakdown and identify the dominant kernel.  Recording ~140 events per step
    # perturbs the step by ~20 % (measured), so inside the timed region only the dominant kernel is bracketed
    # (two events per step).
    probe = 0 if args.no_kernel_timers else min(2, args.warmup)
    for _ in range(args.warmup - probe):
        loss = step.step(g)
    kt_all = {}
    if probe:
        sync()
        ops.KernelTimer.start()
        for _ in range(probe):
            loss = step.step(g)
        kt_all = ops.KernelTimer.stop()
    hip_keys = [k for k in kt_all if k[0] not in ("gemm_nt", "gemm_tn", "absmax")]
    dom = max(hip_keys, key=lambda k: sum(kt_all[k])) if hip_keys else None           # dominant HBM-bound kernel
    dom_all = max(kt_all, key=lambda k: sum(kt_all[k])) if kt_all else None            # dominant kernel of the step
    bracket = [k for k in (dom, dom_all) if k is not None]
    sync()
    # Timed region.  Default: every step is a replay of the captured step (two HIP graphs around the gradient
    # all-reduce, TrainStep.capture) - eagerly the ~230 launches and autograd's host work per step take the host as
    # long as the GPU needs for the kernels (7.5 ms), so a slow host core would be what is measured.  --eager times
    # eagerly issued steps.  The work per step is identical.
    launch, capture_error = "eager", None
    if not args.eager:
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
    sync()
    if launch == "eager" and bracket:
        ops.KernelTimer.start(only=bracket)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run_step()
    sync()
    elapsed = time.perf_counter() - t0
    kt_dom = ops.KernelTimer.stop() if (bracket and launch == "eager") else {}
    eager_leg = None
    if (launch != "eager" or not kt_all) and not args.no_eager_leg and not args.no_kernel_timers:
        # a replay cannot carry HIP events per launch: the dominant kernels' launches are bracketed in eagerly
        # issued steps right after the timed region (same kernels, same operands, same stream)
        n_leg = min(args.steps, 20)
        ops.KernelTimer.start(only=bracket or None)      # no instrumented warm-up steps (--warmup 0): time every kernel here
        t1 = time.perf_counter()
        for _ in range(n_leg):
            step.step(g)
        sync()
        eager_leg = {"ms_per_step": (time.perf_counter() - t1) / n_leg * 1e3, "steps": n_leg}
        kt_dom = ops.KernelTimer.stop()
        if not kt_all:
            kt_all, probe = dict(kt_dom), n_leg
            hip_keys = [k for k in kt_all if k[0] not in ("gemm_nt", "gemm_tn", "absmax")]
            dom = max(hip_keys, key=lambda k: sum(kt_all[k])) if hip_keys else None
            dom_all = max(kt_all, key=lambda k: sum(kt_all[k])) if kt_all else None
    kt = dict(kt_all)
    loss_val = float(loss)

    tot = torch.tensor([elapsed, float(E), float(N)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, E_all, N_all = float(mx[0]), float(sm[1]), float(sm[2])
    else:
        E_all, N_all = float(E), float(N)

    if rank == 0:
        L = cfg.CONV_LAYERS
        ms = elapsed / args.steps * 1e3
        value = E_all * L * args.steps / elapsed
        out = {
            "metric": "message-passing edges/sec (fwd+bwd), batched trees", "value": value, "unit": "layer-edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config} training step (fwd+loss+bwd+allreduce+SGD), {args.trees} trees/GPU, "
                                   f"random fan-out trees n~U[120,180], fp32, dropout {'off' if args.no_dropout else 'on'}",
                       "trees_per_gpu": args.trees, "global_trees": args.trees * world, "nodes": int(N_all),
                       "edges": int(E_all), "conv_layers": L, "trainable_params": n_params,
                       "parallelism": f"dp{world}", "launch": launch,
                       "gemm": ("split-fp16 x3 MFMA, fp32 accumulate (fp32-GEMM accuracy)" if ops.GEMM_MODE == "f16x3"
                                else "fp32 (rocBLAS/hipBLASLt via torch.mm)")},
            "graph_edges_per_s": E_all * args.steps / elapsed, "loss": loss_val,
        }
        if kt:
            nprobe = max(probe, 1)
            gemm_keys = [k for k in kt if k[0] in ("gemm_nt", "gemm_tn", "absmax")]
            gemm_kt = {k: kt.pop(k) for k in gemm_keys}
            if gemm_kt:
                fl = sum(2.0 * k[1] * k[2] * k[3] * len(v) for k, v in gemm_kt.items() if k[0] != "absmax") / nprobe
                g_ms = sum(sum(v) for k, v in gemm_kt.items() if k[0] != "absmax") / nprobe
                out["gemm"] = {"kernel": "spgnn_gemm_nt/tn (split-fp16, 3 MFMA products, fp32 accumulate)",
                               "ms_per_step": g_ms, "fp32_equiv_TFLOPs": fl / (g_ms * 1e-3) / 1e12,
                               "mfma_f16_TFLOPs": 3 * fl / (g_ms * 1e-3) / 1e12, "mfma_f16_peak_TFLOPs": 2500.0,
                               "frac_of_f16_peak": 3 * fl / (g_ms * 1e-3) / 1e12 / 2500.0,
                               "absmax_ms_per_step": sum(sum(v) for k, v in gemm_kt.items() if k[0] == "absmax") / nprobe,
                               # per shape (M, N, K): launches per step, mean ms, fp32-equivalent TFLOP/s
                               "per_shape": {"_".join(str(x) for x in k): [len(v) // nprobe, round(sum(v) / len(v), 4),
                                                                           round(2.0 * k[1] * k[2] * k[3] / (sum(v) / len(v) * 1e-3) / 1e12, 1)]
                                             for k, v in sorted(gemm_kt.items(), key=lambda kv: -sum(kv[1])) if k[0] != "absmax"},
                               "measured_in": f"{nprobe} instrumented warm-up step(s)"}
            agg = {k: (sum(v) / len(v), sum(v), len(v)) for k, v in kt.items()}
            mp_ms = sum(t for _, t, _ in agg.values()) / nprobe
            where = ("timed region" if (launch == "eager" and eager_leg is None) else
                     "eagerly issued launches right after the timed region") + \
                    " (HIP events on the launch stream around every launch of this kernel)"
            traffic_tab = {}
            tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(tpath):
                try:
                    traffic_tab = json.load(open(tpath))
                except Exception:
                    traffic_tab = {}

            def hbm_roofline(key):
                times = kt_dom.get(key) or kt_all[key]
                avg_ms = sum(times) / len(times)
                bytes_alg = algorithmic_bytes(key)
                ach = bytes_alg / (avg_ms * 1e-3) / 1e9
                return {"bound": "hbm", "kernel": key[0], "shape": list(key[1:]), "achieved": ach, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic_tab.get("_".join(str(x) for x in key)),
                        "algorithmic_bytes_per_launch": bytes_alg, "avg_launch_ms": avg_ms, "launches": len(times),
                        "measured_in": where}

            def mfma_roofline(key):
                # algorithmic flops of one launch: 2 M N K (an fp32 product); the kernel executes them as three fp16
                # MFMA products (hi*hi + hi*lo + lo*hi), so the matrix pipe does 3x that: `achieved` counts the
                # EXECUTED fp16 MFMA flops against the dense fp16 peak; `algorithmic_TFLOPs` is the fp32 product rate.
                times = kt_dom.get(key) or kt_all[key]
                avg_ms = sum(times) / len(times)
                fl = 2.0 * key[1] * key[2] * key[3]
                ach = 3.0 * fl / (avg_ms * 1e-3) / 1e12
                return {"bound": "mfma", "kernel": "spgnn_" + key[0], "shape": list(key[1:]), "achieved": ach, "peak": 2500.0,
                        "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": traffic_tab.get("_".join(str(x) for x in key)),
                        "algorithmic_flops_per_launch": fl, "executed_mfma_flops_per_launch": 3.0 * fl,
                        "algorithmic_TFLOPs": fl / (avg_ms * 1e-3) / 1e12, "fp32_matrix_peak_TFLOPs": 157.3,
                        "avg_launch_ms": avg_ms, "launches": len(times), "measured_in": where,
                        "note": "fp32 operands split on the fly into fp16 hi+lo, three MFMA products, fp32 accumulate"}

            if dom_all is not None and dom_all[0] in ("gemm_nt", "gemm_tn"):
                out["roofline"] = mfma_roofline(dom_all)          # the step's dominant kernel
                if dom is not None:
                    out["roofline_hbm"] = hbm_roofline(dom)        # and its dominant HBM-bound kernel
            elif dom is not None:
                out["roofline"] = hbm_roofline(dom)
            mp_bytes = sum(algorithmic_bytes(k) * n for k, (_, _, n) in agg.items()) / nprobe
            # SURVEY.md 8d's K1-K3 (scores + softmax + aggregation and their backward passes) on their own: the kernels
            # the >= 50 % HBM-roofline target is stated over (its 2.43 G layer-edges/s roofline for st_pgat_spgnn_3 at 8 TB/s)
            gk = {k: v for k, v in agg.items() if k[0].startswith(("gat_fwd", "gat_bwd", "gat_agg"))}
            gat_part = None
            if gk:
                g_ms = sum(t for _, t, _ in gk.values()) / nprobe
                g_bytes = sum(algorithmic_bytes(k) * n for k, (_, _, n) in gk.items()) / nprobe
                gat_part = {"ms_per_step": g_ms, "algorithmic_GB_per_step": g_bytes / 1e9,
                            "achieved_GBps": g_bytes / (g_ms * 1e-3) / 1e9,
                            "frac_of_hbm_peak": g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                            "layer_edges_per_s": E * L / (g_ms * 1e-3)}
            out["message_passing"] = {"ms_per_step": mp_ms, "share_of_step": mp_ms / ms,
                                      "algorithmic_GB_per_step": mp_bytes / 1e9,
                                      "achieved_GBps": mp_bytes / (mp_ms * 1e-3) / 1e9,
                                      "frac_of_hbm_peak": mp_bytes / (mp_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                      "layer_edges_per_s_mp_only": E * L / (mp_ms * 1e-3),
                                      "gat_kernels": gat_part,
                                      "measured_in": f"{nprobe} instrumented warm-up step(s)",
                                      "per_kernel_ms": {"_".join(str(x) for x in k): round(a, 5) for k, (a, _, _) in sorted(agg.items())}}
        if capture_error:
            out["config"]["capture_error"] = capture_error
        if eager_leg:
            eager_leg["value"] = E_all * L / (eager_leg["ms_per_step"] * 1e-3) if world == 1 else None
            eager_leg["note"] = "same step issued eagerly after the timed region (host-paced when the host is slower than the GPU)"
            out["eager"] = eager_leg
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, model, samples, min(args.cpu_trees, args.trees))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
