"""Shared helpers for the parity tests."""
import numpy as np
import torch


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """Normwise relative error max|a-b| / max|b| (the 1e-5 bar of BASELINE.json.north_star)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    denom = b.abs().max().item()
    return (a - b).abs().max().item() / (denom if denom > 0 else 1.0)


def mixed_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """Elementwise mixed error max_i |a_i - b_i| / (|b_i| + rms(b)): a value of e means every element satisfies
    |a - b| <= e * |b| + e * rms(b) - small elements are held to the tensor's typical magnitude, not to its maximum
    (the normwise ``rel_err`` says nothing about them)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    rms = b.pow(2).mean().sqrt().item()
    return ((a - b).abs() / (b.abs() + (rms if rms > 0 else 1.0))).max().item()


def keep_scale_host(seed: int, idx: np.ndarray, p: float) -> np.ndarray:
    """Bit-exact host copy of the kernels' counter-based attention-dropout mask
    (spgnn_kernels.hip keep_scale): returns 1/(1-p) where kept, 0 where dropped."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx.astype(np.uint64) + np.uint64(1))) & M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.where(u >= np.float32(p), np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0)).astype(np.float32)


def keep4_scale_host(seed: int, N: int, widths, p: float) -> np.ndarray:
    """Host copy of spgnn_cat_dropout's mask for sources of the given widths: one 64-bit hash per group of four
    columns of a source, 16 bits per element."""
    total = int(sum(widths))
    out = np.zeros((N, total), dtype=np.float32)
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    thr = np.uint64(int(np.float32(p) * np.float32(65536.0)))
    rows = np.arange(N, dtype=np.int64)[:, None]
    off = 0
    for w in widths:
        c = np.arange(0, w, 4, dtype=np.int64)[None, :]
        idx = (rows * total + off + c).astype(np.uint64)
        with np.errstate(over="ignore"):
            z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(1))) & M
            z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M
            z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M
            z = z ^ (z >> np.uint64(31))
        for j in range(4):
            cols = off + c[0] + j
            ok = cols < off + w
            bits = (z >> np.uint64(16 * j)) & np.uint64(0xFFFF)
            out[:, cols[ok]] = np.where(bits[:, ok] >= thr, np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0))
        off += w
    return out


def tree_batch_edges(ns, seed=0):
    from spgnn_amd import synthetic
    from spgnn_amd.graph import edges_from_adj
    rng = np.random.default_rng(seed)
    srcs, dsts, off = [], [], 0
    for n in ns:
        u, v = edges_from_adj(synthetic.random_tree_adj(n, rng))
        srcs.append(u + off); dsts.append(v + off); off += n
    return np.concatenate(srcs), np.concatenate(dsts), off


def mix64_host(seed: int, idx: np.ndarray) -> np.ndarray:
    """Host copy of the kernels' counter hash (spgnn_kernels.hip mix64)."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx.astype(np.uint64) + np.uint64(1))) & M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M
        return z ^ (z >> np.uint64(31))


def sample_neighbors_host(indptr, indices, eid, seeds, fanout, seed):
    """Bit-exact host copy of spgnn_sample_neighbors (selection sampling over the CSC slots of every seed):
    returns (src parent ids, parent edge ids, per-seed counts), edges grouped by seed."""
    src, eids, cnt = [], [], []
    for v in seeds:
        b, d = int(indptr[v]), int(indptr[v + 1] - indptr[v])
        k = d if fanout is None or fanout < 0 or fanout > d else fanout
        r = (mix64_host(seed, np.arange(b, b + d, dtype=np.int64)) >> np.uint64(32)).astype(np.uint64)
        m = 0
        for i in range(d):
            if m >= k:
                break
            left, need = d - i, k - m
            take = need >= left or int((int(r[i]) * left) >> 32) < need
            if take:
                src.append(int(indices[b + i])); eids.append(int(eid[b + i])); m += 1
        cnt.append(m)
    return np.array(src, np.int64), np.array(eids, np.int64), np.array(cnt, np.int64)
