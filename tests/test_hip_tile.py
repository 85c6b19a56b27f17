"""Tree-resident LDS tile kernels (csrc/spgnn_tile.hip; BASELINE.json north_star: "LDS staging of neighbour tiles") against the
row kernels of spgnn_kernels.hip they stand in for, through the same C ABI wrappers (ops.gat_fwd_raw / gat_bwd_raw and the
bf16 layer).  The forward and the source-major half do the row kernels' arithmetic in the row kernels' order: bit-identical;
the destination-major half sums its per-edge dots over another lane geometry: fp32 rounding.  Tiles that split trees
(neighbours outside the tile: the global-load path) must give the same results as closed tiles.  Oracle parity of the layers
and models that now run on tiles is what tests/test_hip_layers.py / test_hip_models.py / test_hip_bf16.py check as before."""
import numpy as np
import pytest
import torch

from spgnn_amd import _capi, ops, ops_bf16, synthetic
from spgnn_amd.graph import TreeGraph
from tests.util import rel_err, tree_batch_edges

pytestmark = pytest.mark.gpu


def _graph(ns, seed=0):
    s, d, n = tree_batch_edges(ns, seed)
    g = TreeGraph((s, d), n)
    g.batch_num_nodes_list = list(ns)
    g.batch_num_edges_list = [3 * k - 2 for k in ns]
    return g.to("cuda"), n


def _inputs(n, H, D, dtype, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    y = (torch.randn(n, 2 * H * D, device="cuda", generator=gen)).to(dtype)          # [ft | res]
    s = torch.randn(n, 2 * H, device="cuda", generator=gen)
    bias = torch.randn(H * D, device="cuda", generator=gen) * 0.1
    g_out = torch.randn(n, H * D, device="cuda", generator=gen).to(dtype)
    return y, s, bias, g_out


def _run_fp32(csc, y, s, bias, g_out, H, D, act, p, od, score):
    HD = H * D
    ft, res = y[:, :HD], y[:, HD:]
    total = HD + 8
    buf = torch.zeros(y.shape[0], total, device="cuda")
    out_drop = (od, 77, total, 4) if od > 0 else None
    out_view = buf[:, 4:4 + HD] if od > 0 else None
    blk = ops.new_scale_block("cuda")
    out, _, attn = ops.gat_fwd_raw(csc, ft, s[:, :H], s[:, H:], res, bias, H, D, 0.2, act, p, 1234, out=out_view, out_drop=out_drop,
                                   out_absmax=blk)
    g_y = torch.zeros_like(y)
    g_s = torch.zeros_like(s)
    blk2 = ops.new_scale_block("cuda")
    al = torch.randn(HD, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) if score else None
    ar = torch.randn(HD, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4)) if score else None
    g_e = ops.gat_bwd_raw(csc, ft, s[:, :H], s[:, H:], attn, g_out, out if act != ops.ACT_NONE else None, H, D, 0.2, act, p, 1234,
                          g_y[:, HD:], g_y[:, :HD], g_s[:, :H], g_s[:, H:], absmax=blk2, score_l=al, score_r=ar, out_drop=out_drop)
    return dict(out=out.clone(), attn=attn, g_y=g_y, g_s=g_s, g_e=g_e, blk=blk.clone(), blk2=blk2.clone())


@pytest.mark.parametrize("H,D", [(2, 64), (2, 128), (1, 64), (1, 128), (4, 64)])
@pytest.mark.parametrize("act,p,od,score", [(ops.ACT_ELU, 0.0, 0.0, False), (ops.ACT_TANH, 0.1, 0.1, True), (ops.ACT_NONE, 0.0, 0.0, True)])
def test_fp32_tile_kernels_equal_the_row_kernels(H, D, act, p, od, score, monkeypatch):
    g, n = _graph([150, 23, 1, 180, 64, 121, 7], seed=H * 100 + D)
    csc = g.csc("cuda")
    monkeypatch.setattr(ops, "TILE_FORCE", True)
    assert all(ops.tile_plan(csc, H, D, 4, kind) is not None for kind in ("fwd", "dst", "src"))
    y, s, bias, g_out = _inputs(n, H, D, torch.float32, seed=5)
    monkeypatch.setattr(ops, "TILE_FORCE", False); monkeypatch.setattr(ops, "TILE_KERNELS", False)
    ref = _run_fp32(csc, y, s, bias, g_out, H, D, act, p, od, score)
    monkeypatch.setattr(ops, "TILE_FORCE", True)
    got = _run_fp32(csc, y, s, bias, g_out, H, D, act, p, od, score)
    HD = H * D
    if H == 4:
        # four heads of 64 columns: the row kernel takes its whole-table softmax (a head narrower than its 64-lane team), whose
        # exp sum runs over the slots in order instead of pairwise: attention equal to fp32 rounding, not to the bit
        assert rel_err(got["out"], ref["out"]) < 1e-6 and rel_err(got["attn"], ref["attn"]) < 1e-6
        assert rel_err(got["g_y"], ref["g_y"]) < 2e-6 and rel_err(got["g_e"], ref["g_e"]) < 2e-6 and rel_err(got["g_s"], ref["g_s"]) < 2e-6
        return
    assert torch.equal(got["out"], ref["out"]) and torch.equal(got["attn"], ref["attn"])          # forward: bit-identical
    assert torch.equal(got["blk"], ref["blk"])                                                     # ... and the same scale maxima
    assert torch.equal(got["g_y"][:, HD:], ref["g_y"][:, HD:])                                     # g_pre: elementwise, bit-identical
    assert rel_err(got["g_e"], ref["g_e"]) < 2e-6 and rel_err(got["g_s"], ref["g_s"]) < 2e-6      # dots in another lane order
    assert rel_err(got["g_y"][:, :HD], ref["g_y"][:, :HD]) < 2e-6                                  # g_ft (score term: g_el, g_er)
    if not score:
        assert torch.equal(got["g_y"][:, :HD], ref["g_y"][:, :HD])                                 # without it: the same sums in the same order


@pytest.mark.parametrize("cap", [16, 48, 100])
def test_tiles_that_split_trees_give_the_same_results(cap, monkeypatch):
    """A tile need not be closed under neighbours: with a tile size below the tree size most tiles have neighbours outside
    (parents / children in the next tile) and take the global-load path for them."""
    H, D = 2, 64
    g, n = _graph([150, 90, 33, 170], seed=9)
    csc = g.csc("cuda")
    y, s, bias, g_out = _inputs(n, H, D, torch.float32, seed=6)
    monkeypatch.setattr(ops, "TILE_FORCE", False); monkeypatch.setattr(ops, "TILE_KERNELS", False)
    ref = _run_fp32(csc, y, s, bias, g_out, H, D, ops.ACT_ELU, 0.1, 0.1, True)
    monkeypatch.setattr(ops, "TILE_FORCE", True)
    monkeypatch.setattr(ops, "TILE_NODES", cap)
    t, n_tiles = csc.tiles(cap)
    tp = t.cpu().numpy()
    assert tp[0] == 0 and tp[n_tiles] == n and (np.diff(tp[:n_tiles + 1]) <= cap).all() and (np.diff(tp) >= 0).all()
    got = _run_fp32(csc, y, s, bias, g_out, H, D, ops.ACT_ELU, 0.1, 0.1, True)
    HD = H * D
    assert torch.equal(got["out"], ref["out"]) and torch.equal(got["attn"], ref["attn"])
    assert torch.equal(got["g_y"][:, HD:], ref["g_y"][:, HD:])
    assert rel_err(got["g_e"], ref["g_e"]) < 2e-6 and rel_err(got["g_s"], ref["g_s"]) < 2e-6
    assert rel_err(got["g_y"][:, :HD], ref["g_y"][:, :HD]) < 2e-6


def test_tile_table_cuts_at_tree_boundaries_and_has_a_fixed_length():
    g, n = _graph([150, 23, 1, 180, 64, 121, 7, 300], seed=1)
    csc = g.csc("cuda")
    t, n_tiles = csc.tiles(192)
    tp = t.cpu().numpy()
    assert len(tp) == 2 * ((n + 191) // 192) + 3
    cuts = set(np.cumsum([0, 150, 23, 1, 180, 64, 121, 7, 300]).tolist())
    used = tp[:n_tiles + 1]
    inside_big = [b for b in used if b not in cuts]
    assert len(inside_big) == 1 and 546 < inside_big[0] < 846                     # only the 300-node tree is split
    assert (tp[n_tiles:] == n).all()                                              # empty tiles behind the used ones


@pytest.mark.parametrize("H,D", [(2, 64), (2, 128), (2, 256)])
@pytest.mark.parametrize("act,drop", [(ops.ACT_ELU, 0.0), (ops.ACT_ELU, 0.1), (ops.ACT_NONE, 0.0)])
def test_bf16_layer_on_tiles_equals_the_row_kernels(H, D, act, drop, monkeypatch):
    """The bf16 GATConv layer of BASELINE config 4 (ops_bf16._GATLayerBf16Fn) with its three traversals on LDS tiles against
    the same layer on the row kernels: forward bit-identical (bf16 rows), gradients to one bf16 rounding of g_ft."""
    BF = torch.bfloat16
    g, n = _graph([150, 23, 180, 64, 121, 7, 1], seed=H + D)
    csc = g.csc("cuda")
    monkeypatch.setattr(ops, "TILE_FORCE", True)
    assert all(ops.tile_plan(csc, H, D, 2, kind) is not None for kind in ("fwd", "dst", "src"))
    K = 128
    torch.manual_seed(D)
    x0 = torch.randn(n, K, device="cuda").to(BF)
    w_fc = (torch.randn(H * D, K, device="cuda") * 0.1)
    w_res = (torch.randn(H * D, K, device="cuda") * 0.1)
    al = torch.randn(1, H, D, device="cuda") * 0.1
    ar = torch.randn(1, H, D, device="cuda") * 0.1
    bias = torch.randn(H * D, device="cuda") * 0.1
    cot = torch.randn(n, H * D, device="cuda").to(BF)
    res = {}
    for tiles in (False, True):
        monkeypatch.setattr(ops, "TILE_KERNELS", tiles)
        monkeypatch.setattr(ops, "TILE_FORCE", tiles)
        ps = [t.clone().requires_grad_(True) for t in (w_fc, w_res, al, ar, bias)]
        x = x0.clone().requires_grad_(True)
        out, attn = ops_bf16._GATLayerBf16Fn.apply(x, ps[0], ps[1], ps[2], ps[3], ps[4], csc, H, D, 0.2, act, drop, 99, False,
                                                   (drop, 55) if drop > 0 else None)
        (out.float() * cot.float()).sum().backward()
        res[tiles] = (out.detach().clone(), attn.detach().clone(), x.grad.clone(), [p.grad.clone() for p in ps])
    o0, a0, gx0, gp0 = res[False]
    o1, a1, gx1, gp1 = res[True]
    assert torch.equal(o0, o1) and torch.equal(a0, a1)
    assert rel_err(gx1, gx0) < 2 ** -7
    for q0, q1 in zip(gp0, gp1):
        assert rel_err(q1, q0) < 2e-3, rel_err(q1, q0)


def test_arena_rewrites_the_tile_table_in_place():
    """A captured step bakes in the tile table's address and the launch grid: the arena's table has its fixed length, is
    rewritten in place for every loaded batch and cuts at THAT batch's tree boundaries."""
    from spgnn_amd.arena import BatchArena
    a = synthetic.make_batch(5, rank=1, device="cuda", pos_enc_dim=None)
    b = synthetic.make_batch(5, rank=2, device="cuda", pos_enc_dim=None)
    if BatchArena.class_key(a, 512) != BatchArena.class_key(b, 512):
        pytest.skip("the two batches fell into different size classes")
    ar = BatchArena(a, 512)
    ag = ar.load(a)
    acsc = ag.csc("cuda")
    t, n_grid = acsc.tiles(192)
    ptr, first = t.data_ptr(), t.cpu().numpy().copy()
    assert n_grid == len(first) - 1                                            # the grid covers the whole fixed-length table
    ag = ar.load(b)
    assert ag.batch_num_nodes_list == list(b.batch_num_nodes_list) + [ar.n_cap - b.number_of_nodes()]     # the pad component last
    t2, n_grid2 = ag.csc("cuda").tiles(192)
    second = t2.cpu().numpy()
    assert t2.data_ptr() == ptr and n_grid2 == n_grid and len(second) == len(first)
    cuts_b = set(np.cumsum([0] + list(b.batch_num_nodes_list)).tolist())
    real = [x for x in second if x <= b.number_of_nodes()]
    assert all(x in cuts_b for x in real)                                      # this batch's boundaries, not the first one's
    assert not np.array_equal(first, second)


def test_the_default_table_puts_only_the_measured_winners_on_tiles():
    """ops.TILE_TABLE (tools/tile_ab.py, profiles/r05_tile_ab_*.json): the source-major half on bf16 rows and on 64-column fp32
    rows, for batches of at least TILE_MIN_NODES nodes; everything else stays on the row kernels."""
    g = synthetic.make_batch(256, rank=0, device="cuda", pos_enc_dim=None, fv_dim=8)
    csc = g.csc("cuda")
    assert csc.num_nodes >= ops.TILE_MIN_NODES
    assert ops.tile_plan(csc, 2, 64, 2, "src") is not None and ops.tile_plan(csc, 2, 256, 2, "src") is not None
    assert ops.tile_plan(csc, 1, 64, 4, "src") is not None and ops.tile_plan(csc, 2, 64, 4, "src") is None
    assert ops.tile_plan(csc, 2, 64, 2, "fwd") is None and ops.tile_plan(csc, 2, 64, 2, "dst") is None
    small = synthetic.make_batch(8, rank=0, device="cuda", pos_enc_dim=None, fv_dim=8).csc("cuda")
    assert ops.tile_plan(small, 2, 64, 2, "src") is None


@pytest.mark.parametrize("name,bf16", [("st_gat_3", False), ("st_gat_6", True)])
def test_captured_arena_step_on_tiles_follows_each_loaded_batch(name, bf16, monkeypatch):
    """End to end what test_arena_rewrites_the_tile_table_in_place checks at table level: a TrainStep captured on batch A with
    all three traversals on LDS tiles, then REPLAYED on batch B (other trees, other boundaries; the arena rewrites the tile
    table in place, the captured grid stays), against eager steps on the row kernels over the same batches.  A stale table
    would put B's nodes into A's tiles (neighbours outside the staged range, rows of other trees): losses and parameters
    would part at once."""
    import copy
    from spgnn_amd import models
    from spgnn_amd.arena import BatchArena
    from spgnn_amd.configs import class_weight_list, get_config
    from spgnn_amd.train import TrainStep
    cfg = get_config(name)
    torch.manual_seed(11)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None)
    model.set_gcn_only()
    if bf16:
        models.set_storage_dtype(model, torch.bfloat16)
    model.eval()                                                                   # no dropout masks: same arithmetic both ways
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    ga = synthetic.make_batch(6, rank=3, device="cuda", pos_enc_dim=None)
    gb = synthetic.make_batch(6, rank=4, device="cuda", pos_enc_dim=None)
    GRAN = 2048
    assert BatchArena.class_key(ga, GRAN) == BatchArena.class_key(gb, GRAN)
    assert list(ga.batch_num_nodes_list) != list(gb.batch_num_nodes_list)
    m_rows = copy.deepcopy(model)
    monkeypatch.setattr(ops, "TILE_KERNELS", True)
    monkeypatch.setattr(ops, "TILE_FORCE", True)
    ts = TrainStep(model, w, 1.0, 1e-3, 0.9)
    la = ts.run_batch(ga, 4, granule=GRAN)
    arena = next(iter(ts._arenas.values()))
    acsc = arena.graph.csc("cuda")
    assert ops.tile_plan(acsc, 2, 64, 2 if bf16 else 4, "src") is not None         # the captured step did run on tiles
    lb = ts.run_batch(gb, 4, granule=GRAN)
    assert len(ts._captures) == 1 and arena.loads == 2
    monkeypatch.setattr(ops, "TILE_KERNELS", False)
    monkeypatch.setattr(ops, "TILE_FORCE", False)
    ts_r = TrainStep(m_rows, w, 1.0, 1e-3, 0.9)
    for _ in range(4):
        lra = ts_r.step(ga)
    for _ in range(4):
        lrb = ts_r.step(gb)
    tol = 2e-2 if bf16 else 1e-5
    n = ts.bucket.numel
    assert rel_err(la, lra) < tol and rel_err(lb, lrb) < tol, (rel_err(la, lra), rel_err(lb, lrb))
    assert rel_err(ts.bucket.flat_param[:n], ts_r.bucket.flat_param[:n]) < tol
