"""the weight-gradient side stream as a CU PARTITION instead of a time slice.

The shipped step issues each level's weight-gradient pair (spgnn_gemm_tn_pair) on a side stream beside the next level's
traversals (ops.SideLaunch).  Here the side stream is created with hipExtStreamCreateWithCUMask so those products may only use
K of the 256 CUs (bit i of the mask -> XCD i % 8: the lowest K bits give K / 8 CUs in every XCD), and the step is timed
eagerly and as a captured HIP graph for K in a sweep, ONE process, interleaved rounds, medians:

  * step time (HIP events around 20 steps);
  * the K1-K3 traversal launches inside eager steps (ops.KernelTimer, events on the launch stream) = the in-step HBM figure;
  * whether the mask survives capture: a graph captured with a masked side stream against one captured with a plain one.

usage: python3 tools/cu_mask_ab.py [trees] [K,K,...]      ->  one JSON line + a table on stderr
"""
import ctypes
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import models, ops, synthetic, train  # noqa: E402
from spgnn_amd.configs import class_weight_list, get_config  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(k_cus: int, total: int = 256) -> torch.cuda.Stream:
    """A stream whose kernels may run on the lowest ``k_cus`` mask bits only (spread evenly over the 8 XCDs)."""
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(k_cus):
        mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask({k_cus}) -> {rc}")
    return torch.cuda.ExternalStream(s.value)


def stream_mask(stream) -> int:
    """CUs the runtime reports for a stream (hipExtStreamGetCUMask), or -1."""
    words = 8
    mask = (ctypes.c_uint32 * words)()
    rc = hip.hipExtStreamGetCUMask(ctypes.c_void_p(stream.cuda_stream), ctypes.c_uint32(words), mask)
    return sum(bin(w).count("1") for w in mask) if rc == 0 else -1


def main():
    trees = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [96, 128, 160, 192]
    cfg = get_config("st_pgat_spgnn_3")
    g = synthetic.make_batch(trees, rank=0, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    g.csc("cuda")
    N = g.number_of_nodes()
    variants = [("off", None)] + ([("plain", 0)] if os.environ.get("NO_PLAIN", "0") != "1" else []) + [(f"K{k}", k) for k in ks if k > 0]
    steps = {}
    for name, k in variants:
        torch.manual_seed(0)
        model = models.build_model(cfg.MODEL).cuda()
        model.init(None); model.set_gcn_only(); model.train(True)
        ops.OVERLAP_TN = k is not None
        st = train.TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, 1e-4, 0.9, seed=1)
        if k:                                   # the side stream of this step object: masked
            st._tn_side = ops.SideLaunch(torch.device("cuda", 0))
            st._tn_side.side = masked_stream(k)
            st._mask_reported = stream_mask(st._tn_side.side)
        for _ in range(5):
            st.step(g)
        torch.cuda.synchronize()
        steps[name] = (st, k)
    ops.OVERLAP_TN = True

    def eager_ms(st, k, n=20):
        ops.OVERLAP_TN = k is not None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            st.step(g)
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    def k123_ms(st, k, n=10):
        """K1-K3 launches inside eager steps of this variant + the TN products themselves."""
        ops.OVERLAP_TN = k is not None
        ops.KernelTimer.start()
        for _ in range(2):
            st.step(g)
        keys = list(ops.KernelTimer.stop())
        gat = [q for q in keys if q[0].startswith(("lspe_", "gat_agg", "gat_fwd", "gat_bwd"))]
        tn = [q for q in keys if q[0].startswith("gemm_tn")]
        ops.KernelTimer.start(only=gat + tn)
        for _ in range(n):
            st.step(g)
        rec = ops.KernelTimer.stop()
        per = {}
        for q, v in rec.items():
            per.setdefault(q[0], 0.0)
            per[q[0]] += sum(v) / n
        return sum(v for q, v in per.items() if not q.startswith("gemm_tn")), sum(v for q, v in per.items() if q.startswith("gemm_tn")), per
    res = {name: {"eager": [], "k123": [], "tn": []} for name in steps}
    for name, (st, k) in steps.items():
        eager_ms(st, k, 5)
    for r in range(5):
        for name, (st, k) in steps.items():
            res[name]["eager"].append(eager_ms(st, k))
            a, b, per = k123_ms(st, k)
            res[name]["k123"].append(a); res[name]["tn"].append(b); res[name]["per"] = per
    # captured: one graph per variant, replays interleaved
    caps = {}
    for name, (st, k) in steps.items():
        ops.OVERLAP_TN = k is not None
        try:
            st.capture(g)
            caps[name] = st
        except Exception as e:
            res[name]["capture_error"] = repr(e)[:200]
    ops.OVERLAP_TN = True

    def replay_ms(st, n=20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            st.replay()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    for st in caps.values():
        replay_ms(st, 5)
    for r in range(7):
        for name, st in caps.items():
            res[name].setdefault("replay", []).append(replay_ms(st))
    med = statistics.median
    out = {"workload": f"st_pgat_spgnn_3, {trees} trees (N={N}), fp32, dropout on", "variants": {}}
    for name, (st, k) in steps.items():
        r = res[name]
        out["variants"][name] = {"side_stream_cus": k, "mask_reported": getattr(st, "_mask_reported", None),
                                 "eager_ms": round(med(r["eager"]), 4), "k123_in_step_ms": round(med(r["k123"]), 4),
                                 "tn_in_step_ms": round(med(r["tn"]), 4), "replay_ms": round(med(r["replay"]), 4) if r.get("replay") else None,
                                 "capture_error": r.get("capture_error"),
                                 "per_kernel_ms": {q: round(v, 4) for q, v in sorted(r["per"].items())}}
        v = out["variants"][name]
        print(f"{name:6s} side CUs {str(k):5s} eager {v['eager_ms']:.3f} ms  replay {v['replay_ms']}  K1-K3 in step {v['k123_in_step_ms']:.3f} ms  TN in step {v['tn_in_step_ms']:.3f} ms",
              file=sys.stderr)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
