"""The split-fp16 MFMA GEMMs (spgnn_gemm_nt / spgnn_gemm_tn) against fp64 products: exact on small-integer
data (catches fragment-layout errors), fp32-GEMM accuracy on random data, ragged M/N/K tails, strided
operands, scaled tensors far outside the fp16 range."""
import pytest
import torch

from spgnn_amd import ops
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _mat(r, c, scale=1.0, ints=False):
    cp = (c + 3) // 4 * 4 + 4                      # row stride > width, 16-byte aligned rows
    buf = torch.randint(-8, 9, (r, cp), device="cuda").float() if ints else torch.randn(r, cp, device="cuda") * scale
    return buf[:, :c]


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 3, 7), (128, 128, 32), (129, 127, 33), (300, 200, 1063), (257, 129, 39),
                                   (1000, 22, 1024), (64, 4096, 192)])
def test_gemm_nt_exact_on_integers_and_accurate_on_random(M, N, K):
    a, b = _mat(M, K, ints=True), _mat(N, K, ints=True)
    assert torch.equal(ops.gemm_nt(a, b), a @ b.t())                     # products and sums exact in fp16/fp32
    a, b = _mat(M, K), _mat(N, K, 0.05)
    ref = a.double() @ b.double().t()
    c = ops.gemm_nt(a, b, ops.pow2_scale(a), ops.pow2_scale(b))
    assert rel_err(c, ref) < max(3 * rel_err(a @ b.t(), ref), 2e-6)      # fp32-GEMM class accuracy


@pytest.mark.parametrize("R,M,N", [(1, 1, 1), (40, 7, 5), (100, 130, 33), (1000, 256, 128), (513, 64, 300), (4097, 1024, 39),
                                   (76410, 256, 256), (16385, 128, 128), (300, 64, 64)])   # incl. empty trailing splits
def test_gemm_tn_exact_on_integers_and_accurate_on_random(R, M, N):
    a, b = _mat(R, M, ints=True), _mat(R, N, ints=True)
    assert torch.equal(ops.gemm_tn(a, b), a.t() @ b)
    a, b = _mat(R, M, 1e-5), _mat(R, N)                                   # gradient-sized values: need the scale
    ref = a.double().t() @ b.double()
    c = ops.gemm_tn(a, b, ops.pow2_scale(a), ops.pow2_scale(b))
    assert rel_err(c, ref) < max(3 * rel_err(a.t() @ b, ref), 2e-6)


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (300, 200, 1063), (513, 257, 39), (700, 512, 64), (256, 256, 96), (257, 1030, 100)])
def test_gemm_nt_256x256_kernel(M, N, K):
    """gemm_nt_f16x3_v3 (256 x 256 tiles, buffer-descriptor loads, one staging set) forced on every shape: bit-identical
    to the 256 x 128 kernel - same split, same products, same summation order per output element - with every epilogue
    option (rank-J update, bias, activation, score partials)."""
    a, b = _mat(M, K), _mat(N, K, 0.05)
    sa, sb = ops.pow2_scale(a), ops.pow2_scale(b)
    u, v = torch.randn(M, 4, device="cuda"), torch.randn(4, (N + 3) // 4 * 4, device="cuda")
    v[:, N:] = 0
    bias = torch.randn(N, device="cuda")
    kw = dict(upd_u=u, upd_v=v, bias=bias, act=ops.ACT_ELU)
    C = N // 64 * 64
    sc = [torch.zeros(M, max(C // 64, 1), 2, device="cuda") for _ in range(2)]
    if C:
        kw.update(score_l=torch.randn(C, device="cuda"), score_r=torch.randn(C, device="cuda"))
    sc.append(torch.zeros_like(sc[0]))
    ref, ref2 = ops.gemm_nt(a, b, sa, sb, tile=4), ops.gemm_nt(a, b, sa, sb, tile=4, **kw, **({"score_out": sc[0]} if C else {}))   # 256 x 128
    out, out2 = ops.gemm_nt(a, b, sa, sb, tile=5), ops.gemm_nt(a, b, sa, sb, tile=5, **kw, **({"score_out": sc[1]} if C else {}))   # 256 x 256
    low, low2 = ops.gemm_nt(a, b, sa, sb, tile=2), ops.gemm_nt(a, b, sa, sb, tile=2, **kw, **({"score_out": sc[2]} if C else {}))   # 128 x 128
    ai, bi = _mat(M, K, ints=True), _mat(N, K, ints=True)
    exact = ops.gemm_nt(ai, bi, tile=5)
    assert torch.equal(out, ref) and torch.equal(out2, ref2) and torch.equal(sc[0], sc[1])
    assert torch.equal(low, ref) and torch.equal(low2, ref2) and torch.equal(sc[0], sc[2])
    assert torch.equal(ops.gemm_nt(a, b, sa, sb), ref)                      # the shape heuristic picks one of the three
    assert torch.equal(exact, ai @ bi.t())


@pytest.mark.parametrize("M,N,K", [(700, 384, 1063), (300, 130, 39), (1030, 1024, 384), (513, 260, 100)])
def test_gemm_nt_presplit_b_is_bit_identical(M, N, K):
    """b_presplit: the weight operand split once by spgnn_presplit (packed fp16 hi / lo pairs in place of every four fp32
    values, ragged widths zero padded) gives bit for bit what the in-kernel conversion gives, in every tile variant; the
    scale derived from block maxima equals pow2_scale; a second matrix (the transpose) rides in the same launch."""
    torch.manual_seed(M + N + K)
    Kp, Mp = (K + 3) // 4 * 4, (N + 3) // 4 * 4
    a = torch.randn(M, Kp + 4, device="cuda")[:, :K]
    b = (torch.randn(N, Kp, device="cuda") / 7)[:, :K]
    bt = b.t().contiguous() if N % 4 == 0 else torch.nn.functional.pad(b.t(), (0, Mp - N)).contiguous()[:, :N]
    sa, sb = ops.pow2_scale(a), ops.pow2_scale(b)
    part = b.abs().amax(1).contiguous()                       # any set of block maxima covering the matrix
    b_ps, bt_ps, sc = ops.presplit(b, partials=part, w2=bt)
    assert torch.equal(sc, sb)
    b_ps2 = ops.presplit(b, scale=sb)[0]
    assert torch.equal(b_ps.view(torch.int32), b_ps2.view(torch.int32))
    ref = ops.gemm_nt(a, b, sa, sb)
    assert rel_err(ref, a.double() @ b.double().t()) < 2e-6
    for tile in (0, 2, 4, 5):
        got = ops.gemm_nt(a, b_ps, sa, sb, tile=tile, b_presplit=True)
        assert torch.equal(got, ref), tile
    g = torch.randn(M, Mp + 4, device="cuda")[:, :N]          # the input-gradient product reads W^T pre-split
    sg = ops.pow2_scale(g)
    assert torch.equal(ops.gemm_nt(g, bt_ps, sg, sb, b_presplit=True), ops.gemm_nt(g, bt, sg, sb))


@pytest.mark.parametrize("M,N,K", [(700, 384, 1063), (300, 130, 39), (1030, 1024, 384), (513, 260, 100), (4200, 512, 1024)])
def test_gemm_presplit_a_is_bit_identical(M, N, K):
    """a_presplit (round 4): a CONSTANT fp32 operand - node data feeding a model's first layer, reference models.py:425-428 -
    split once per loader batch by spgnn_presplit.  The forward product (A pre-split next to pre-split weights, every tile
    variant) and the weight-gradient product (X pre-split as the TN kernel's B operand, single and pair launches, ragged
    widths and row counts) give bit for bit what the in-kernel conversion gives."""
    torch.manual_seed(M * 3 + N + K)
    Kp = (K + 3) // 4 * 4
    x = torch.randn(M, Kp + 4, device="cuda")[:, :K]                       # padded row stride, as cat_padded leaves it
    w = (torch.randn(N, Kp, device="cuda") / 7)[:, :K]
    sx, sw = ops.pow2_scale(x), ops.pow2_scale(w)
    x_ps, w_ps = ops.presplit(x, scale=sx)[0], ops.presplit(w, scale=sw)[0]
    assert x_ps.stride() == x.stride()
    ref = ops.gemm_nt(x, w, sx, sw)
    for tile in (0, 2, 4, 5):
        assert torch.equal(ops.gemm_nt(x_ps, w_ps, sx, sw, tile=tile, b_presplit=True, a_presplit=True), ref), tile
    with pytest.raises(RuntimeError):                                       # A-only is not instantiated: the ABI says so
        ops.gemm_nt(x_ps, w, sx, sw, a_presplit=True)
    # weight gradient g^T x with x pre-split (the TN kernel's B operand)
    Np = (N + 3) // 4 * 4
    g = torch.randn(M, Np + 4, device="cuda")[:, :N]
    sg = ops.pow2_scale(g)
    gw, cs = ops.gemm_tn(g, x, sg, sx, want_colsum=True)
    gw2, cs2 = ops.gemm_tn(g, x_ps, sg, sx, want_colsum=True, b_presplit=True)
    assert torch.equal(gw, gw2) and torch.equal(cs, cs2)
    assert rel_err(gw, g.double().t() @ x.double()) < 2e-6
    # pair launches: both products with pre-split constants (a level's structure + position layers on node data)
    x1 = torch.randn(M, 40, device="cuda")[:, :39]
    w1 = (torch.randn(96, 40, device="cuda") / 3)[:, :39]
    s1, sw1 = ops.pow2_scale(x1), ops.pow2_scale(w1)
    x1_ps, w1_ps = ops.presplit(x1, scale=s1)[0], ops.presplit(w1, scale=sw1)[0]
    r0, r1 = ops.gemm_nt(x, w, sx, sw), ops.gemm_nt(x1, w1, s1, sw1)
    p0 = ops.NtProblem(x_ps, w_ps, sx, sw, b_presplit=True, a_presplit=True)
    p1 = ops.NtProblem(x1_ps, w1_ps, s1, sw1, b_presplit=True, a_presplit=True)
    o0, o1 = ops.gemm_nt_pair(p0, p1)
    assert torch.equal(o0, r0) and torch.equal(o1, r1)
    g1 = torch.randn(M, 96, device="cuda")
    sg1 = ops.pow2_scale(g1)
    t_ref = (ops.gemm_tn(g, x, sg, sx), ops.gemm_tn(g1, x1, sg1, s1))
    t0 = ops.TnProblem(g, x_ps, sg, sx, b_presplit=True)
    t1 = ops.TnProblem(g1, x1_ps, sg1, s1, b_presplit=True)
    q0, q1 = ops.gemm_tn_pair(t0, t1)
    assert torch.equal(q0, t_ref[0]) and torch.equal(q1, t_ref[1])


@pytest.mark.parametrize("R,M,N", [(5000, 512, 300), (3001, 256, 39), (20000, 1024, 1063), (4100, 384, 200), (2500, 128, 128), (70, 256, 64)])
def test_gemm_tn_256_row_tiles_are_bit_identical(R, M, N):
    """Round 4: the weight-gradient kernel in 256 x 128 block tiles (8 waves, one workgroup per CU) - same split ranges, same
    accumulation order per output element as the 128 x 128 form: bit-identical results and column sums, ragged M / N / R, X
    pre-split or not, single and pair launches (the second product of a pair runs in the first one's tile)."""
    torch.manual_seed(R + M + N)
    Np = (N + 3) // 4 * 4
    g = torch.randn(R, M, device="cuda")
    x = torch.randn(R, Np + 4, device="cuda")[:, :N]
    sg, sx = ops.pow2_scale(g), ops.pow2_scale(x)
    x_ps = ops.presplit(x, scale=sx)[0]
    for splits in (1, 5):
        ref, cs = ops.gemm_tn(g, x, sg, sx, want_colsum=True, tile=128, splits=splits)
        got, cs2 = ops.gemm_tn(g, x, sg, sx, want_colsum=True, tile=256, splits=splits)
        assert torch.equal(ref, got) and torch.equal(cs, cs2), splits
        assert torch.equal(ops.gemm_tn(g, x_ps, sg, sx, tile=256, splits=splits, b_presplit=True), ref)
    assert rel_err(ref, g.double().t() @ x.double()) < 2e-6
    auto = ops.TnProblem(g, x, sg, sx)
    assert auto.rows == (256 if (M % 256 == 0 and R >= 4096 and M * N >= 384 * 1024) else 128)
    # pair: a second, small product rides in the first one's tile shape
    g1 = torch.randn(R, 96, device="cuda")
    x1 = torch.randn(R, 40, device="cuda")[:, :39]
    s1, t1 = ops.pow2_scale(g1), ops.pow2_scale(x1)
    want = (ops.gemm_tn(g, x, sg, sx, tile=128, splits=3), ops.gemm_tn(g1, x1, s1, t1, tile=128, splits=2))
    for tile in (128, 256):
        a, b = ops.gemm_tn_pair(ops.TnProblem(g, x, sg, sx, tile=tile, splits=3), ops.TnProblem(g1, x1, s1, t1, tile=tile, splits=2))
        assert torch.equal(a, want[0]) and torch.equal(b, want[1]), tile


def test_const_operand_is_built_once_and_refreshed_in_place():
    """ops.const_operand: the pre-split image of a marked batch constant is made on first use and reused while the tensor is
    unchanged; refresh_batch_constant rewrites scale and image IN PLACE after the data was overwritten (batch arena)."""
    x = torch.randn(600, 64, device="cuda")
    sx = ops.operand_scale(x)
    assert ops.const_operand(x, sx) == (x, False)                           # not marked: nothing happens
    ops.mark_batch_constant(x)
    ps, on = ops.const_operand(x, sx)
    assert on and ops.const_operand(x, sx)[0] is ps
    assert torch.equal(ps.view(torch.int32), ops.presplit(x, scale=sx)[0].view(torch.int32))
    x.copy_(torch.randn(600, 64, device="cuda") * 37.0)
    ops.refresh_batch_constant(x)
    assert ops.operand_scale(x) is sx and torch.equal(sx, ops.pow2_scale(x))
    ps2, on = ops.const_operand(x, sx)
    assert on and ps2 is ps and torch.equal(ps.view(torch.int32), ops.presplit(x, scale=ops.pow2_scale(x))[0].view(torch.int32))


def test_gemm_scaling_keeps_extreme_magnitudes():
    """Values far outside fp16's range (1e-9 .. 1e+7) survive through the power-of-two scales."""
    for mag_a, mag_b in [(1e-9, 1e-3), (1e7, 1e3), (1e-12, 1e6)]:
        a, b = _mat(200, 96, mag_a), _mat(64, 96, mag_b)
        sa, sb = ops.pow2_scale(a), ops.pow2_scale(b)
        assert float(sa) == 2.0 ** round(torch.log2(sa).item()) and torch.isfinite(sa)
        c = ops.gemm_nt(a, b, sa, sb)
        assert torch.isfinite(c).all() and rel_err(c, a.double() @ b.double().t()) < 2e-6
    z = torch.zeros(8, 8, device="cuda")
    assert float(ops.pow2_scale(z)) == 1.0                                # all-zero tensor: neutral scale
    assert torch.equal(ops.gemm_nt(z, z, ops.pow2_scale(z), ops.pow2_scale(z)), z)


@pytest.mark.parametrize("ratio,body_bound", [(1e4, 3e-6), (1e6, 2e-5)])
def test_gemm_heavy_tailed_operand_error_on_body_rows(ratio, body_bound):
    """One per-tensor power-of-two scale serves the whole operand: hi + lo keeps 22 bits only for elements within ~2^17 of
    the tensor maximum; below that lo falls into fp16's subnormals (absolute spacing 2^-24 after scaling) and an element
    2^k below the maximum keeps about 38 - k bits.  A plausible late-training gradient: a body ~N(0, 1e-3) with 0.01 %
    entries `ratio` times larger.  Rows of A that hold no outlier are compared with fp64, relative to the magnitude of
    THOSE rows' results.  Measured envelope: outliers 1e4 x the body - fp32-class (an fp32 GEMM over
    K = 512 terms sits at ~1e-6); 1e6 x - 9e-6, i.e. one decimal digit lost on the body rows, still inside the 1e-5 bar;
    the normwise error of the whole product is unaffected.  (Per-row-block scales would remove the effect for the NT
    kernel only - the weight-gradient kernel reduces over the rows; not built.  Round 4 adds the GUARD instead: operands
    whose rows / blocks leave the 2^18 envelope are flagged on the device and counted, test_range_monitor_* below.)"""
    M, N, K = 4096, 256, 512
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randn(M, K, device="cuda", generator=g) * 1e-3
    hot = torch.rand(M, K, device="cuda", generator=g) < 1e-4
    a = torch.where(hot, a * ratio, a)
    b = torch.randn(N, K, device="cuda", generator=g) * 0.05
    body = ~hot.any(dim=1)
    assert 0.3 < float(body.float().mean()) < 0.99
    c = ops.gemm_nt(a, b, ops.pow2_scale(a), ops.pow2_scale(b))
    ref = a.double() @ b.double().t()
    assert rel_err(c, ref) < 2e-6                                           # normwise over the whole result: unaffected
    assert rel_err(c[body], ref[body]) < body_bound                         # rows that never touch an outlier
    # weight gradient: A^T X with the heavy-tailed tensor as the (R, M) operand; columns of A without an outlier
    x = torch.randn(M, 192, device="cuda", generator=g)
    gw = ops.gemm_tn(a, x, ops.pow2_scale(a), ops.pow2_scale(x))
    refw = a.double().t() @ x.double()
    cols = ~hot.any(dim=0)
    assert rel_err(gw, refw) < 2e-6
    if bool(cols.any()):
        assert rel_err(gw[cols], refw[cols]) < body_bound


@pytest.mark.parametrize("ratio", [1e6, 1e8])
def test_wide_range_form_keeps_body_rows_at_fp32_accuracy(ratio, monkeypatch):
    """SPGNN_GEMM_WIDE / SPGNN_TN_WIDE (round 4): lo kept as 2^11 lo and the cross products in a second
    accumulator set.  The heavy-tailed operand of the test above - 0.01 % of the entries `ratio` times the body - now gives
    rows (NT) and columns (TN) that never touch an outlier at 1e-5 of THEIR OWN magnitude, at 1e6 x and 1e8 x; on ordinary
    data the wide form agrees with the narrow one to fp32 rounding, is exact on integers, and its pre-split images
    (spgnn_presplit wide = 1) are consumed bit-identically to the in-kernel conversion."""
    M, N, K = 4096, 256, 512
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randn(M, K, device="cuda", generator=g) * 1e-3
    hot = torch.rand(M, K, device="cuda", generator=g) < 1e-4
    a = torch.where(hot, a * ratio, a)
    b = torch.randn(N, K, device="cuda", generator=g) * 0.05
    body = ~hot.any(dim=1)
    sa, sb = ops.pow2_scale(a), ops.pow2_scale(b)
    ref = a.double() @ b.double().t()
    narrow = ops.gemm_nt(a, b, sa, sb, wide=False)
    wide = ops.gemm_nt(a, b, sa, sb, wide=True)
    e_n, e_w = rel_err(narrow[body], ref[body]), rel_err(wide[body], ref[body])
    print(f"ratio {ratio:g}: body rows narrow {e_n:.2e}, wide {e_w:.2e}")
    assert e_w < 1e-5 and rel_err(wide, ref) < 2e-6
    monkeypatch.setattr(ops, "GEMM_WIDE", True)              # presplit / TnProblem read the module switch
    x = torch.randn(M, 192, device="cuda", generator=g)
    sx = ops.pow2_scale(x)
    gw = ops.gemm_tn(a, x, sa, sx)
    refw = a.double().t() @ x.double()
    cols = ~hot.any(dim=0)
    assert rel_err(gw, refw) < 2e-6
    if bool(cols.any()):
        assert rel_err(gw[cols], refw[cols]) < 1e-5
    # pre-split images in the wide form: bit-identical to the in-kernel wide conversion, NT (both operands) and TN
    a_ps, b_ps, x_ps = ops.presplit(a, scale=sa)[0], ops.presplit(b, scale=sb)[0], ops.presplit(x, scale=sx)[0]
    assert torch.equal(ops.gemm_nt(a_ps, b_ps, sa, sb, b_presplit=True, a_presplit=True), wide)
    assert torch.equal(ops.gemm_tn(a, x_ps, sa, sx, b_presplit=True), gw)
    assert not torch.equal(b_ps.view(torch.int32), ops.presplit(b, scale=sb, wide=False)[0].view(torch.int32))
    # ordinary operands: wide == narrow to fp32 rounding; exact on integers
    p_, q_ = torch.randn(700, 300, device="cuda"), torch.randn(130, 300, device="cuda")
    sp, sq = ops.pow2_scale(p_), ops.pow2_scale(q_)
    assert rel_err(ops.gemm_nt(p_, q_, sp, sq), ops.gemm_nt(p_, q_, sp, sq, wide=False)) < 1e-6
    pi, qi = _mat(300, 96, ints=True), _mat(64, 96, ints=True)
    assert torch.equal(ops.gemm_nt(pi, qi), pi @ qi.t())
    with pytest.raises(RuntimeError):                        # the wide form has 128 x 128 tiles only
        ops.gemm_nt(p_, q_, sp, sq, tile=5)


def test_range_monitor_flags_rows_outside_the_envelope():
    """Round 4 guard for the per-tensor scale: a product whose slot-block operand holds whole rows more
    than 2^18 below the tensor maximum - where hi + lo no longer carries 22 bits - sets the block's range flag; the next
    spgnn_step_begin adds the flagged blocks to the pool's device counter and re-arms them.  A well-conditioned operand leaves
    the flag clear.  (Detection only: the result itself is the enveloped one the test above measures.)"""
    dev = torch.device("cuda")
    pool = ops.scale_pool(dev)
    before = ops.range_violations(dev)
    pool.begin()
    try:
        M, K, N = 4096, 256, 128
        w = torch.randn(N, K, device=dev) / 9
        sw = ops.pow2_scale(w)
        # rows 0..15 are 1e7 times the body: every slot but sixteen sits outside the 2^18 = 2.6e5 envelope
        a_bad = torch.randn(M, K, device=dev) * 1e-3
        a_bad[:16] *= 1e7
        a_ok = torch.randn(M, K, device=dev)
        a_ok[:16] *= 1e3                                                 # inside the envelope: no flag
        blocks = []
        for a in (a_bad, a_ok):
            # a producer that folds row maxima into the slots: the fused concat / dropout kernel with p = 0
            y = ops.cat_dropout((a,), 0.0, 0)
            blk = y._spgnn_scale[1]
            assert not ops.range_flag(blk)
            c = ops.gemm_nt(y, w, blk, sw)
            assert torch.isfinite(c).all()
            blocks.append(blk)
        torch.cuda.synchronize()
        assert ops.range_flag(blocks[0]) and not ops.range_flag(blocks[1])
        # the weight-gradient kernel reads the same block as its A operand's scale
        y_bad = ops.cat_dropout((a_bad,), 0.0, 0)
        assert not ops.range_flag(y_bad._spgnn_scale[1])
        g = ops.gemm_tn(y_bad, a_ok, y_bad._spgnn_scale[1], ops.pow2_scale(a_ok))
        torch.cuda.synchronize()
        assert torch.isfinite(g).all() and ops.range_flag(y_bad._spgnn_scale[1])
    finally:
        pool.end()
    pool.begin(); pool.end()                                            # the next step's first launch harvests the flags
    assert ops.range_violations(dev) == before + 2


def test_gemm_mode_switch_gives_same_layer_output(monkeypatch):
    import torch.nn.functional as F
    from spgnn_amd import nn as snn
    from spgnn_amd.graph import TreeGraph
    from tests.util import tree_batch_edges
    s, d, n = tree_batch_edges([150, 99], 3)
    g = TreeGraph((s, d), n).to("cuda")
    torch.manual_seed(0)
    layer = snn.GATConv(1063, 256, 2, residual=True, activation=F.elu).cuda()
    x = ops.cat_padded((torch.randn(n, 1024, device="cuda"), torch.rand(n, 39, device="cuda")))
    outs = {}
    for mode in ("f16x3", "fp32"):
        monkeypatch.setattr(ops, "GEMM_MODE", mode)
        outs[mode] = layer(g, x)
    assert rel_err(outs["f16x3"], outs["fp32"]) < 5e-6


@pytest.mark.parametrize("M,N,K,J", [(300, 200, 64, 4), (5000, 768, 512, 4), (4097, 39, 256, 2), (129, 1063, 33, 16)])
def test_gemm_nt_fused_rank_update(M, N, K, J):
    """C = A B^T + U V (the score-gradient term of the input gradient) applied in the epilogue, exact fp32."""
    a, b = _mat(M, K), _mat(N, K, 0.05)
    u = torch.randn(M, J, device="cuda")
    npad = (N + 15) // 16 * 16
    v = torch.zeros(J, npad, device="cuda"); v[:, :N] = torch.randn(J, N, device="cuda")
    out = torch.empty(M, (N + 3) // 4 * 4, device="cuda")[:, :N]
    ops.gemm_nt(a, b, ops.pow2_scale(a), ops.pow2_scale(b), out=out, upd_u=u, upd_v=v)
    ref = a.double() @ b.double().t() + u.double() @ v[:, :N].double()
    assert rel_err(out, ref) < 2e-6


@pytest.mark.parametrize("R,M,N", [(1000, 256, 128), (4097, 1024, 39), (76410, 512, 384), (513, 130, 300)])
def test_gemm_tn_column_sums(R, M, N):
    a, b = _mat(R, M, 1e-4), _mat(R, N)
    c, cs = ops.gemm_tn(a, b, ops.pow2_scale(a), ops.pow2_scale(b), want_colsum=True)
    assert rel_err(c, a.double().t() @ b.double()) < 2e-6
    assert rel_err(cs, a.double().sum(0)) < 2e-6 and cs.shape == (M,)


@pytest.mark.parametrize("N,K,C,act", [(5000, 1024, 256, ops.ACT_NONE), (1000, 100, 36, ops.ACT_RELU), (700, 64, 1024, ops.ACT_ELU),
                                        (2048, 33, 50, ops.ACT_TANH), (600, 40, 34, ops.ACT_NONE)])
def test_linear_on_matrix_cores_matches_torch(N, K, C, act):
    """ops.linear (nn.Linear / weight products of GraphConv, GINConv, SAGEConv on the split-fp16 GEMMs) against
    torch in fp64: output and all three gradients, with odd widths (row padding paths) and fused activations."""
    import torch.nn.functional as F
    torch.manual_seed(N + K + C)
    x = torch.randn(N, K, device="cuda", requires_grad=True)
    w = (torch.randn(C, K, device="cuda") / K ** 0.5).requires_grad_(True)
    b = torch.randn(C, device="cuda", requires_grad=True)
    ops.KernelTimer.start()
    y = ops.linear(x, w, b, act)
    cot = torch.randn(N, C, device="cuda")
    (y * cot).sum().backward()
    used = {k[0] for k in ops.KernelTimer.stop()}
    assert "gemm_nt" in used and "gemm_tn" in used
    f = {ops.ACT_NONE: lambda t: t, ops.ACT_ELU: F.elu, ops.ACT_TANH: torch.tanh, ops.ACT_RELU: torch.relu}[act]
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = f(F.linear(xd, wd, bd))
    (yd * cot.double()).sum().backward()
    y32 = f(F.linear(x.detach(), w.detach(), b.detach()))
    tol = max(3 * rel_err(y32, yd), 2e-6)
    assert rel_err(y, yd) < tol
    for got, want in ((x.grad, xd.grad), (w.grad, wd.grad), (b.grad, bd.grad)):
        assert rel_err(got, want) < 5e-6
    wt = torch.randn(K, C, device="cuda", requires_grad=True)                       # GraphConv's (in, out) weight
    y2 = ops.linear(x.detach(), wt.t())
    y2.sum().backward()
    assert rel_err(y2, x.detach().double() @ wt.detach().double()) < 2e-6 and wt.grad.shape == (K, C)


@pytest.mark.parametrize("S,shape", [(2, (8, 12)), (7, (130, 36)), (32, (64, 68)), (128, (22, 1024)), (1000, (4, 512)), (3, (5, 7))])
def test_sum_partials(S, shape):
    """spgnn_sum_partials (the split-K partial-tile reduction) against an fp64 sum; run-to-run bitwise reproducible."""
    part = torch.randn((S,) + shape, device="cuda")
    out = ops.sum_partials(part)
    ref = part.double().sum(0)
    assert out.shape == ref.shape and rel_err(out, ref) < 1e-6
    assert torch.equal(out, ops.sum_partials(part))


@pytest.mark.parametrize("R1,R2,K", [(512, 512, 1063), (256, 0, 39), (64, 64, 128), (5, 3, 7), (1024, 1024, 192)])
def test_weight_cat(R1, R2, K):
    """spgnn_weight_cat: [w_a; w_b] with 16-byte rows, its transpose, its GEMM scale - against cat / t() / pow2_scale -
    and the row-range views its autograd backward hands to the two parameters."""
    wa = torch.randn(R1, K, device="cuda", requires_grad=True)
    wb = torch.randn(R2, K, device="cuda", requires_grad=True) if R2 else None
    w = ops.weight_cat(wa, wb, want_t=True)
    ref = torch.cat([wa, wb], 0) if R2 else wa
    assert torch.equal(w, ref.detach()) and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0
    buf = w._base if w._base is not None else w
    assert float(buf[:, K:].abs().sum()) == 0.0                                  # pad columns are zero
    wt = ops._tagged(w, "_spgnn_t")
    assert ops._tagged(w, "_spgnn_ps") is not None and ops._tagged(w, "_spgnn_t_ps") is not None
    assert torch.equal(wt, ref.detach().t()) and wt.stride(0) % 4 == 0
    assert float(ops.operand_scale(w)) == float(ops.pow2_scale(ref.detach().contiguous() if K % 4 == 0 else
                                                               torch.nn.functional.pad(ref.detach(), (0, -K % 4))))
    g = torch.randn(R1 + R2, K, device="cuda")
    w.backward(g)
    assert torch.equal(wa.grad, g[:R1]) and (wb is None or torch.equal(wb.grad, g[R1:]))
    # an in-place write to the operand invalidates what weight_cat attached (the split form and the scale describe old values)
    with torch.no_grad():
        w.mul_(2.0)
    assert ops._tagged(w, "_spgnn_ps") is None and ops._tagged(w, "_spgnn_t_ps") is None and ops._b_operand(w)[1] is False


@pytest.mark.parametrize("M,N,K", [(1000, 1024, 384), (257, 128, 96), (4100, 512, 64)])
def test_gemm_nt_headmean(M, N, K):
    """spgnn_gemm_nt_headmean: the second head's product also writes 0.5 * (head 0 + head 1); C and the mean are
    bit-identical to spgnn_gemm_nt followed by the mean, on interior and edge tiles, with bias + ELU."""
    a, b = _mat(M, K), _mat(N, K, 0.05)
    sa, sb = ops.pow2_scale(a), ops.pow2_scale(b)
    bias = torch.randn(N, device="cuda")
    other = torch.randn(M, N, device="cuda")
    ref = ops.gemm_nt(a, b, sa, sb, bias=bias, act=ops.ACT_ELU)
    out, mean = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.gemm_nt_headmean(a, b, sa, sb, out, other, mean, bias=bias, act=ops.ACT_ELU)
    assert torch.equal(out, ref)
    assert torch.equal(mean, 0.5 * (ref + other))


def test_weight_prep_batched_equals_per_layer():
    """spgnn_weight_prep (every layer's operands in one call) against spgnn_weight_cat + spgnn_presplit per layer: [W_fc; W_res]
    with 16-byte rows, its transpose, both pre-split forms and the scale, bit for bit; shapes of st_pgat_spgnn_3 plus ragged ones."""
    torch.manual_seed(0)
    shapes = [(512, 512, 1063, False), (256, 256, 39, False), (256, 256, 768, True), (128, 128, 256, True), (64, 0, 128, True),
              (5, 3, 7, True), (33, 31, 70, False)]
    params = [(torch.randn(r1, k, device="cuda") * (0.05 + i), torch.randn(r2, k, device="cuda") if r2 else None, t)
              for i, (r1, r2, k, t) in enumerate(shapes)]
    ref = []
    for a, b, t in params:
        w = ops.weight_cat(a, b, want_t=t)
        ref.append((w.clone(), ops.operand_scale(w).clone(), ops._tagged(w, "_spgnn_ps").clone(),
                    ops._tagged(w, "_spgnn_t").clone() if t else None, ops._tagged(w, "_spgnn_t_ps").clone() if t else None))
    for rep in range(2):                                       # the second pass reuses the cached buffers and table
        with ops.prepared_weights(params):
            assert len(ops._PREP_ACTIVE) == len(params)
            for (a, b, t), (w0, s0, ps0, t0, tps0) in zip(params, ref):
                w = ops.weight_cat(a, b, want_t=t)
                assert torch.equal(w, w0) and float(ops.operand_scale(w)) == float(s0)
                assert torch.equal(ops._tagged(w, "_spgnn_ps").view(torch.int32), ps0.view(torch.int32))
                if t:
                    assert torch.equal(ops._tagged(w, "_spgnn_t"), t0)
                    assert torch.equal(ops._tagged(w, "_spgnn_t_ps").view(torch.int32), tps0.view(torch.int32))
                base = w._base if w._base is not None else w
                assert float(base[:, w.shape[1]:].abs().sum()) == 0.0          # pad columns are zero
        assert not ops._PREP_ACTIVE
    assert len(ops._PREP_CACHE) >= 1


def test_weight_prep_column_mode_feeds_the_aggregate_first_layer():
    """spgnn_weight_prep mode 1: [W_fc | W_res] (the aggregate-first layer's per-head operand), its scale, pre-split form and
    transpose - and a model step that takes them from the prepared pass equals, bit for bit, the step that assembles them
    with torch ops."""
    torch.manual_seed(2)
    H, D, F_ = 2, 64, 24
    wa, wb = torch.randn(H * D, F_, device="cuda") * 0.3, torch.randn(H * D, F_, device="cuda") * 0.1
    for b in (wb, None):
        with ops.prepared_weights([(wa, b, True, "cols")]):
            (dst, ps, dst_t, ps_t, scale, meta), want_t = ops._PREP_ACTIVE[(id(wa), id(b) if b is not None else 0, "cols")]
            ref = torch.cat([wa, b], dim=1) if b is not None else wa
            K = ref.shape[1]
            assert meta == (H * D, 0 if b is None else H * D, K, H * D) and want_t
            assert torch.equal(dst[:, :K], ref) and torch.equal(dst_t[:K, :H * D], ref.t())
            s0 = ops.pow2_scale(ref)
            assert float(scale) == float(s0)
            assert torch.equal(ps[:, :K].contiguous().view(torch.int32), ops.presplit(ref.contiguous(), scale=s0)[0].view(torch.int32))
            assert torch.equal(ps_t[:K, :H * D].contiguous().view(torch.int32),
                               ops.presplit(ref.t().contiguous(), scale=s0)[0].view(torch.int32))
    from spgnn_amd import models, synthetic
    from spgnn_amd.configs import get_config
    cfg = get_config("st_pgat_spgnn_3")
    g = synthetic.make_batch(3, rank=4, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    model = models.build_model(cfg.MODEL).cuda().eval()
    outs = []
    for batch in (True, False):
        ops.BATCH_WEIGHT_PREP = batch
        try:
            model.zero_grad(set_to_none=True)
            logits = model(g)[0]
            logits.square().sum().backward()
            outs.append((logits.detach().clone(), [p.grad.clone() for p in model.parameters()]))
        finally:
            ops.BATCH_WEIGHT_PREP = True
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,shapes", [(9641, ((1024, 1063), (512, 39))), (9641, ((512, 768), (256, 256))), (3000, ((256, 384), (128, 128))),
                                      (20000, ((1024, 1063), (512, 39))), (777, ((130, 70), (64, 8)))])
def test_pair_launches_equal_two_launches(M, shapes):
    """spgnn_gemm_nt_pair / spgnn_gemm_tn_pair (a level's structure and position products in one launch) against two
    launches, bit for bit: forward form with score partials and pre-split weights, and the weight-gradient form with column sums
    and deferred partial sums; the second product runs in the first one's block tile (256 x 256 at M = 20 000)."""
    torch.manual_seed(M)
    xs = [_mat(M, (k + 3) // 4 * 4)[:, :k] for _, k in shapes]
    ws = [_mat(n, (k + 3) // 4 * 4, 0.05)[:, :k] for n, k in shapes]
    sx, sw = [ops.pow2_scale(x) for x in xs], [ops.pow2_scale(w) for w in ws]
    ps = [ops.presplit(w, scale=s_)[0] for w, s_ in zip(ws, sw)]
    vec = [(torch.randn(n // 64 * 64, device="cuda"), torch.randn(n // 64 * 64, device="cuda")) for n, _ in shapes]

    def problems(presplit):
        out = []
        for i, (n, k) in enumerate(shapes):
            sc = n // 64 * 64
            pt = torch.zeros((M, sc // 64, 2), device="cuda") if sc else None
            out.append(ops.NtProblem(xs[i], ps[i] if presplit else ws[i], sx[i], sw[i], score_l=vec[i][0] if sc else None,
                                     score_r=vec[i][1] if sc else None, score_out=pt, b_presplit=presplit))
        return out
    for presplit in (True, False):
        a = problems(presplit); b = problems(presplit)
        ya = ops.gemm_nt_pair(a[0], a[1])
        yb = (b[0].run(), b[1].run())
        for i in range(2):
            assert torch.equal(ya[i], yb[i])
            if a[i].kw["score_out"] is not None:
                assert torch.equal(a[i].kw["score_out"], b[i].kw["score_out"])
    # weight gradients: g (M, n)^T x (M, k)
    gs = [_mat(M, n) for n, _ in shapes]
    sg = [ops.pow2_scale(g_) for g_ in gs]
    res = []
    for pair in (True, False):
        jobs = ops.SumJobs("cuda")
        t = [ops.TnProblem(gs[i], xs[i], sg[i], sx[i], want_colsum=(i == 0), defer=jobs) for i in range(2)]
        r = ops.gemm_tn_pair(t[0], t[1]) if pair else (t[0].launch().finish(), t[1].launch().finish())
        jobs.flush()
        res.append((r[0][0].clone(), r[0][1].clone(), r[1].clone()))
    for x, y in zip(*res):
        assert torch.equal(x, y)
    ref = gs[0].double().t() @ xs[0].double()
    assert rel_err(res[0][0], ref) < 3e-6 and rel_err(res[0][1], gs[0].double().sum(0)) < 1e-5


def test_gemm_takes_scale_blocks():
    """A GEMM operand's scale as a SCALE BLOCK (include/spgnn_hip.h): {-256, 0, 0, 0, m_1 .. m_256} whose largest slot the
    kernel turns into the power-of-two scale itself - bit-identical to passing the scalar scale, for every tile variant of
    the NT product and for the TN product; and ops.ScalePool hands out re-armed blocks in a fixed order."""
    torch.manual_seed(3)
    a, b = _mat(1000, 200), _mat(512, 200, 0.05)
    sa, sb = ops.pow2_scale(a), ops.pow2_scale(b)
    blk = ops.new_scale_block("cuda")
    assert blk.numel() == 260 and float(blk[0]) == -256.0 and float(blk[1:].abs().sum()) == 0.0
    rows = a.abs().amax(1)
    blk[ops.SCALE_HEADER + 7] = rows.max(); blk[ops.SCALE_HEADER + 200] = rows.min()      # any slots, any order
    assert ops.scale_value(blk) == float(sa)
    ref = ops.gemm_nt(a, b, sa, sb)
    for tile in (0, 2, 4, 5):
        assert torch.equal(ops.gemm_nt(a, b, blk, sb, tile=tile), ref), tile
    g = _mat(1000, 128, 1e-3)
    sg = ops.pow2_scale(g)
    gb = ops.new_scale_block("cuda"); gb[ops.SCALE_HEADER] = g.abs().max()
    assert torch.equal(ops.gemm_tn(g, a, gb, blk), ops.gemm_tn(g, a, sg, sa))
    pool = ops.scale_pool("cuda")
    pool.begin()
    b0, b1 = ops.new_scale_block("cuda"), ops.new_scale_block("cuda")
    b0[ops.SCALE_HEADER + 1] = 5.0
    pool.end()
    assert ops.new_scale_block("cuda").data_ptr() not in (b0.data_ptr(), b1.data_ptr())     # outside a step: a block of its own
    pool.begin()
    c0 = ops.new_scale_block("cuda")
    pool.end()
    assert c0.data_ptr() == b0.data_ptr() and float(c0[ops.SCALE_HEADER:].abs().sum()) == 0.0 and float(c0[0]) == -256.0


def test_deferred_partial_sums_equal_single_launches():
    """ops.SumJobs (spgnn_sum_partials_multi): a weight gradient with column sums, one split into two outputs, an
    attention-vector gradient (block diagonal) and a plain skinny gradient reduced in ONE launch - bit-identical to the four
    separate reductions."""
    torch.manual_seed(5)
    R = 20000
    g, x = _mat(R, 512, 1e-3), _mat(R, 200)
    sg, sx = ops.pow2_scale(g), ops.pow2_scale(x)
    ref_w, ref_cs = ops.gemm_tn(g, x, sg, sx, want_colsum=True)
    o1, o2 = torch.empty(512, 120, device="cuda"), torch.empty(512, 80, device="cuda")
    ops.gemm_tn(g, x, sg, sx, out=o1, out2=o2)
    gs, ft = torch.randn(R, 4, device="cuda"), _mat(R, 256)
    ref_bd = ops.scores_bwd_w(gs, ft, blockdiag_heads=2)
    g22, x1 = torch.randn(R, 22, device="cuda"), _mat(R, 1024)
    ref_pl = ops.scores_bwd_w(g22, x1)
    jobs = ops.SumJobs(torch.device("cuda", 0))
    w, cs = ops.gemm_tn(g, x, sg, sx, want_colsum=True, defer=jobs)
    p1, p2 = torch.empty(512, 120, device="cuda"), torch.empty(512, 80, device="cuda")
    ops.gemm_tn(g, x, sg, sx, out=p1, out2=p2, defer=jobs)
    bd = ops.scores_bwd_w(gs, ft, blockdiag_heads=2, defer=jobs)
    pl = ops.scores_bwd_w(g22, x1, defer=jobs)
    assert len(jobs.jobs) == 4
    jobs.flush()
    assert not jobs.jobs
    assert torch.equal(w, ref_w) and torch.equal(cs, ref_cs) and torch.equal(p1, o1) and torch.equal(p2, o2)
    assert torch.equal(bd, ref_bd) and torch.equal(pl, ref_pl)


@pytest.mark.gpu
def test_twenty_deferred_partial_sums_in_one_launch():
    """spgnn_sum_partials_multi takes up to 24 jobs since ABI 53 (a training step queues the reductions of its whole backward
    pass, ops.StepSums): twenty reductions of all three kinds in one launch equal the single launches bit for bit, and a
    twenty-fifth is refused by the library."""
    from spgnn_amd import _capi
    torch.manual_seed(6)
    R = 6000
    refs, outs = [], []
    jobs = ops.SumJobs(torch.device("cuda", 0), local=True)
    assert jobs.MAX == 24
    for i in range(20):
        if i % 3 == 0:
            g, x = _mat(R, 128 + 64 * (i % 2), 1e-3), _mat(R, 100 + 4 * i)
            refs.append(ops.gemm_tn(g, x, want_colsum=True))
            outs.append(ops.gemm_tn(g, x, want_colsum=True, defer=jobs))
        elif i % 3 == 1:
            gs, ft = torch.randn(R, 4, device="cuda"), _mat(R, 128)
            refs.append(ops.scores_bwd_w(gs, ft, blockdiag_heads=2))
            outs.append(ops.scores_bwd_w(gs, ft, blockdiag_heads=2, defer=jobs))
        else:
            g22, x1 = torch.randn(R, 22, device="cuda"), _mat(R, 256)
            refs.append(ops.scores_bwd_w(g22, x1))
            outs.append(ops.scores_bwd_w(g22, x1, defer=jobs))
    assert len(jobs.jobs) == 20
    jobs.flush()
    for r, o in zip(refs, outs):
        for a, b in zip(r if isinstance(r, tuple) else (r,), o if isinstance(o, tuple) else (o,)):
            assert torch.equal(a, b)
    arr = (_capi.SumJob * 25)()
    assert _capi.load().spgnn_sum_partials_multi(arr, 25, ops._stream(refs[-1])) != 0


@pytest.mark.parametrize("M,K,N", [(150, 1063, 1024), (300, 384, 1024), (7, 39, 512), (1, 64, 64), (128, 768, 512), (33, 20, 70), (640, 192, 2048)])
def test_skinny_product_is_an_fp32_gemm(M, K, N):
    """spgnn_gemm_nt_skinny (per-scan inference, reference job_runner.py:2046-2052: M = 100-300 rows): exact fp32 products on the
    fp32 matrix pipe - as close to fp64 as rocBLAS' SGEMM, exact on small-integer data, every ragged edge (rows, columns, K)."""
    from spgnn_amd import ops
    torch.manual_seed(M + K + N)
    Kp = (K + 3) // 4 * 4
    a = torch.zeros(M, Kp, device="cuda")[:, :K]
    b = torch.zeros(N, Kp, device="cuda")[:, :K]
    a.copy_(torch.randn(M, K)); b.copy_(torch.randn(N, K) * 0.1)
    ref = a.double() @ b.double().t()
    got = ops.gemm_nt_skinny(a, b)
    e_lib = float((got.double() - ref).abs().max() / ref.abs().max())
    e_blas = float(((a @ b.t()).double() - ref).abs().max() / ref.abs().max())
    assert e_lib < max(2 * e_blas, 2e-6), (e_lib, e_blas)
    ai, bi = torch.randint(-8, 9, (M, K), device="cuda").float(), torch.randint(-8, 9, (N, K), device="cuda").float()
    a.copy_(ai); b.copy_(bi)
    assert torch.equal(ops.gemm_nt_skinny(a, b), ai @ bi.t())                  # small integers: exact


def test_skinny_product_epilogues():
    """bias + activation, and GATConv's score partials per 64-column block, against torch."""
    from spgnn_amd import ops
    torch.manual_seed(5)
    M, K, N, C = 150, 384, 1024, 512
    a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.1
    bias = torch.randn(N, device="cuda")
    got = ops.gemm_nt_skinny(a, b, bias=bias, act=ops.ACT_ELU)
    ref = torch.nn.functional.elu(a.double() @ b.double().t() + bias.double())
    assert float((got.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    al, ar = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    parts = torch.zeros(M, C // 64, 2, device="cuda")
    y = ops.gemm_nt_skinny(a, b, score_l=al, score_r=ar, score_out=parts)
    yd = (a.double() @ b.double().t())[:, :C].view(M, C // 64, 64)
    rl = (yd * al.double().view(C // 64, 64)).sum(-1)
    rr = (yd * ar.double().view(C // 64, 64)).sum(-1)
    assert float((parts[..., 0].double() - rl).abs().max() / rl.abs().max()) < 2e-6
    assert float((parts[..., 1].double() - rr).abs().max() / rr.abs().max()) < 2e-6
    assert float((y.double() - a.double() @ b.double().t()).abs().max()) < 1e-4


def test_small_inference_batches_take_the_skinny_kernel_and_training_does_not(monkeypatch):
    """ops.skinny_rows: no-grad products of up to SKINNY_ROWS rows; anything with autograd on keeps the split-fp16 kernels (one
    arithmetic for a step's forward and backward)."""
    from spgnn_amd import ops
    a, b = torch.randn(100, 64, device="cuda"), torch.randn(128, 64, device="cuda")
    seen = []
    real = ops.gemm_nt_skinny
    monkeypatch.setattr(ops, "gemm_nt_skinny", lambda *x, **k: (seen.append(1), real(*x, **k))[1])
    with torch.no_grad(), ops.inference_scope():             # what a model call under no_grad sets up (models._InferenceScoped)
        y0 = ops.gemm_nt(a, b)
    assert seen == [1]
    with ops.inference_scope():                               # autograd on at the call: not inference
        y1 = ops.gemm_nt(a, b)
    assert seen == [1] and float((y0 - y1).abs().max() / y1.abs().max()) < 3e-6
    big = torch.randn(ops.SKINNY_ROWS + 1, 64, device="cuda")
    with torch.no_grad(), ops.inference_scope():
        ops.gemm_nt(big, b)
    assert seen == [1] and not ops.INFERENCE
