"""GPU parity tests, kernel family by kernel family: every C-ABI entry (through the ctypes host layer) against the
stock-PyTorch CPU statement of the same ATen op the reference uses.  fp32; tolerances are written in each test."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    import ops as _ops
    _ops.lib()           # raises if libcorrif_gfx950.so is missing: no fall-back
    return _ops


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def cl(t):      # NCDHW -> channels-last dense
    return t.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(t):
    return t.permute(0, 4, 1, 2, 3)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CONVS = [
    # name, Ci, Co, k, stride, pad, replicate, (B, D, H, W), bias
    ("gemm_1x1", 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, (2, 3, 14, 14), False),
    ("1x1_ktail_184", 184, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, (2, 8, 8, 8), True),
    ("1x1_stride2", 64, 96, (1, 1, 1), (1, 2, 2), (0, 0, 0), False, (2, 3, 14, 14), False),
    ("1x3x3", 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), False, (2, 3, 14, 14), False),
    ("1x3x3_stride2", 32, 48, (1, 3, 3), (1, 2, 2), (0, 1, 1), False, (2, 3, 14, 14), False),
    ("1x3x3_stride2_odd", 32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), False, (1, 2, 7, 7), False),
    ("3x3x3_zero", 24, 24, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, (2, 4, 9, 10), True),
    ("3x3x3_rep_small_n", 32, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, (2, 8, 8, 8), True),
    ("3x3x3_rep_320", 320, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, (1, 6, 6, 6), True),
    ("3x3x3_rep_16_8", 16, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, (2, 9, 8, 10), True),
    ("3x3x3_zero_16_16", 16, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, (1, 7, 9, 8), True),
    ("3x3x3_rep_64_32", 64, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, (1, 5, 6, 18), True),
    ("3x3x3_zero_d1", 16, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, (2, 1, 9, 20), True),
    ("3x3x3_rep_d2", 8, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, (2, 2, 17, 33), False),
    ("3x3x3_zero_48", 48, 48, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, (1, 4, 10, 10), True),
    # > 512 tiles: persistent patch workgroups walk several tiles each (resident weights / per-chunk weights / single chunk)
    ("3x3x3_rep_32_8_many_tiles", 32, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, (2, 20, 64, 64), True),
    ("3x3x3_zero_64_16_many_tiles", 64, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, (2, 18, 64, 60), True),
    ("3x3x3_rep_8_32_many_tiles", 8, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, (2, 20, 64, 64), False),
    # replicate padding with n % (4, 4, 16) == 0: the data gradient folds the padding adjoint in its epilogue (every tile touches a border)
    ("3x3x3_rep_16_8_fold", 16, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, (2, 4, 8, 16), True),
    ("3x3x3_rep_8_24_fold", 8, 24, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, (1, 8, 4, 32), False),
    # input channel counts that are multiples of 8 but not of 16 (the two halves of d1_c2 on the compact skip path): 8-channel chunks
    # in the patch weight-gradient kernel (4 column groups), also on a shallow grid and over many tiles
    ("3x3x3_rep_8_8", 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, (2, 8, 8, 32), True),
    ("3x3x3_rep_24_8_many_tiles", 24, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, (2, 12, 64, 64), False),
    ("3x3x3_rep_24_16_shallow", 24, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, (1, 2, 20, 24), True),
    ("1x1_12_to_4", 12, 4, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, (2, 5, 6, 7), True),
    ("1x1_16_16", 16, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, (2, 7, 9, 11), True),
    ("1x1_n8", 8, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, (2, 6, 6, 6), True),
    ("1x1_wide", 2048, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, (1, 4, 4, 4), True),
]


@pytest.mark.parametrize("cfg", CONVS, ids=[c[0] for c in CONVS])
def test_conv3d(ops, cfg):
    name, Ci, Co, k, stride, pad, rep, (B, D, Hh, W), bias = cfg
    x = rnd(B, Ci, D, Hh, W, seed=1)
    w = rnd(Co, Ci, *k, seed=2, scale=1.0 / math.sqrt(Ci * k[0] * k[1] * k[2]))
    b = rnd(Co, seed=3) if bias else None
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    if rep:
        yr = F.conv3d(F.pad(xr, (1, 1, 1, 1, 1, 1), mode="replicate"), wr, br, stride)
    else:
        yr = F.conv3d(xr, wr, br, stride, pad)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy)
    xg = cl(x).to(DEV).requires_grad_()
    wg = w.to(DEV).requires_grad_()
    bg = b.to(DEV).requires_grad_() if bias else None
    yg = ops.conv3d(xg, wg, bg, stride, pad, rep)
    yg.backward(cl(gy).to(DEV))
    torch.cuda.synchronize()
    assert rel(ncdhw(yg), yr) < 2e-6, "forward"
    assert rel(ncdhw(xg.grad), xr.grad) < 2e-6, "data gradient"
    assert rel(wg.grad, wr.grad) < 2e-6, "weight gradient"
    if bias:
        assert rel(bg.grad, br.grad) < 2e-6, "bias gradient"


SPLIT_CASES = [
    # name, M, N, K, b_layout: shapes whose tile grid fills the chip unevenly -> stream-K split (persistent workgroups + fixed-order fixup)
    ("e4_conv_like_392_tiles", 25088, 256, 1152, 0),
    ("e5_like_196_tiles_long_k", 6272, 512, 2304, 0),
    ("few_tiles_long_k_KN_operand", 1024, 384, 4096, 1),
    ("ragged_edges", 3000, 200, 2000, 0),
]


@pytest.mark.parametrize("cfg", SPLIT_CASES, ids=[c[0] for c in SPLIT_CASES])
def test_gemm_stream_k_split(ops, cfg):
    """The stream-K launch (equal shares of the (tile, K tile) space per persistent workgroup, partial tiles summed in a fixed order by
    the fixup kernel) against an fp64 matmul; it must actually be split (workspace > 0), be bit-identical run to run, and agree with
    the one-workgroup-per-tile launch of the same kernel to fp32 rounding (the K sum is associated differently)."""
    import corrif_hip as H
    _, M, N, K, bl = cfg
    A = rnd(M, K, seed=11).to(DEV)
    Bm = (rnd(N, K, seed=12) / math.sqrt(K)).to(DEV)
    Bop = Bm.t().contiguous() if bl else Bm
    bias = rnd(N, seed=13).to(DEV)
    ref = (A.double() @ Bm.double().t() + bias.double()).cpu()

    def run():
        C = torch.empty(M, N, device=DEV)
        ops.gemm(A.data_ptr(), K, Bop.data_ptr(), N if bl else K, bl, C.data_ptr(), N, M, N, K, K, H.gemm_geom(), bias=bias.data_ptr())
        return C

    g = H.Gemm()
    g.A, g.lda, g.Cs, g.B, g.ldb, g.b_layout, g.C, g.ldc = A.data_ptr(), K, K, Bop.data_ptr(), (N if bl else K), bl, A.data_ptr(), N
    g.M, g.N, g.K, g.Z, g.Zi, g.g = M, N, K, 1, 1, H.gemm_geom()
    assert H.lib().corrif_gemm_fwd_workspace(g) > 0, "this shape is expected to take the stream-K path"
    was = ops.STREAM_K
    try:
        ops.STREAM_K = True
        c1, c2 = run(), run()
        ops.STREAM_K = False
        c0 = run()
    finally:
        ops.STREAM_K = was
    torch.cuda.synchronize()
    assert torch.equal(c1, c2)
    assert rel(c1, ref) < 2e-6 and rel(c0, ref) < 2e-6
    assert rel(c1, c0) < 1e-6


@pytest.mark.parametrize("kind", ["1x1", "1x3x3"])
def test_batch_norm_backward_statistics_from_dgrad_epilogue(ops, kind, monkeypatch):
    """conv -> BatchNorm(train) -> ReLU -> conv: the second convolution's data-gradient GEMM epilogue produces (sum g', sum g' xhat) of the
    BatchNorm backward (CorrifGemm.bstats_*, corrif_norm_bwd_pre); same gradients as with the separate reduction pass and as stock ATen."""
    B, D, Hh, W, Ci, Cm, Co = 2, 3, 10, 12, 32, 48, 64
    x = rnd(B, Ci, D, Hh, W, seed=1)
    w1 = rnd(Cm, Ci, 1, 1, 1, seed=2, scale=0.2)
    k2 = (1, 1, 1) if kind == "1x1" else (1, 3, 3)
    p2 = (0, 0, 0) if kind == "1x1" else (0, 1, 1)
    w2 = rnd(Co, Cm, *k2, seed=3, scale=0.1)
    gamma, beta = rnd(Cm, seed=4).abs() + 0.5, rnd(Cm, seed=5)
    go = rnd(B, Co, D, Hh, W, seed=6)
    ref = [t.clone().requires_grad_() for t in (x, w1, w2, gamma, beta)]
    h = F.relu(F.batch_norm(F.conv3d(ref[0], ref[1]), None, None, ref[3], ref[4], True, 0.1, 1e-5))
    F.conv3d(h, ref[2], None, 1, p2).backward(go)

    def run(on):
        monkeypatch.setattr(ops, "BWD_STATS", on)
        for k in ops.NORM_BWD_STATS:
            ops.NORM_BWD_STATS[k] = 0
        t = [cl(x).to(DEV).requires_grad_(), w1.to(DEV).requires_grad_(), w2.to(DEV).requires_grad_(), gamma.to(DEV).requires_grad_(),
             beta.to(DEV).requires_grad_()]
        rm, rv = torch.zeros(Cm, device=DEV), torch.ones(Cm, device=DEV)
        link = {}
        y = ops.batch_norm(ops.conv3d(t[0], t[1]), t[3], t[4], rm, rv, relu_out=True, training=True, bwd_link=link)
        out = ops.conv3d(y, t[2], None, (1, 1, 1), p2, bwd_stats=link)
        out.backward(cl(go).to(DEV))
        torch.cuda.synchronize()
        return [t[0].grad, t[1].grad, t[2].grad, t[3].grad, t[4].grad], dict(ops.NORM_BWD_STATS)

    g_on, n_on = run(True)
    g_off, n_off = run(False)
    assert n_on == {"epilogue": 1, "pass": 0} and n_off == {"epilogue": 0, "pass": 1}
    refs = [ref[0].grad, ref[1].grad, ref[2].grad, ref[3].grad, ref[4].grad]
    for i, (a, b, r) in enumerate(zip(g_on, g_off, refs)):
        a = ncdhw(a) if i == 0 else a
        b = ncdhw(b) if i == 0 else b
        assert rel(a, r) < 2e-5, i
        assert rel(a, b) < 2e-6, i


GROUPED = [
    # name, Ci, Co, k, stride, pad, (B per group, D, H, W), bias, zin, zout
    ("1x1_stack", 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 3, 14, 14), False, "stack", "stack"),
    ("1x1_ragged_rows", 128, 96, (1, 1, 1), (1, 1, 1), (0, 0, 0), (3, 2, 7, 9), False, "stack", "stack"),
    ("1x3x3", 32, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 3, 14, 14), False, "stack", "stack"),
    ("1x3x3_stride2", 32, 48, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 3, 14, 14), False, "stack", "stack"),
    ("1x1_stride2", 64, 96, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 3, 14, 14), False, "stack", "stack"),
    ("adapt_to_cat_small_n", 64, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 3, 14, 14), True, "stack", "cat"),
    ("adapt_to_cat", 256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 7, 7), True, "stack", "cat"),
    ("conv6_ktail_184", 184, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 8, 8, 8), True, "stack", "cat"),
    ("encode_from_cat", 64, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 8, 8, 8), True, "cat", "stack"),
]


@pytest.mark.parametrize("cfg", GROUPED, ids=[c[0] for c in GROUPED])
def test_grouped_conv_equals_its_twins(ops, cfg):
    """ops.conv3d_grouped (the three modality encoders' twin layers as ONE launch per GEMM, mmvit4.py:442-447) against F.conv3d per
    group on the CPU, forward / data gradient / weight gradient / bias gradient, for every layer type and activation layout the
    encoder uses: stacked along the batch axis, or side by side in a concat buffer."""
    _, Ci, Co, k, stride, pad, (B, D, Hh, W), bias, zin, zout = cfg
    G = 3
    xs = [rnd(B, Ci, D, Hh, W, seed=40 + g) for g in range(G)]
    ws = [rnd(Co, Ci, *k, seed=50 + g, scale=(Ci * k[0] * k[1] * k[2]) ** -0.5) for g in range(G)]
    bs = [rnd(Co, seed=60 + g) if bias else None for g in range(G)]
    refs = []
    for g in range(G):
        x, w = xs[g].clone().requires_grad_(), ws[g].clone().requires_grad_()
        b = bs[g].clone().requires_grad_() if bias else None
        refs.append((x, w, b, F.conv3d(x, w, b, stride, pad)))
    gys = [rnd(*refs[g][3].shape, seed=70 + g) for g in range(G)]
    for g in range(G):
        refs[g][3].backward(gys[g])

    def pack(ts, mode):      # NCDHW per group -> channels-last "stack" / "cat" layout
        ts = [cl(t) for t in ts]
        return torch.cat(ts, 0 if mode == "stack" else -1).contiguous()

    def unpack(t, mode, g):
        n = t.shape[0] // G if mode == "stack" else t.shape[-1] // G
        return ncdhw(t[g * n:(g + 1) * n] if mode == "stack" else t[..., g * n:(g + 1) * n])

    xg = pack(xs, zin).to(DEV).requires_grad_()
    wg = [w.to(DEV).requires_grad_() for w in ws]
    bg = [b.to(DEV).requires_grad_() if bias else None for b in bs]
    y = ops.conv3d_grouped(xg, wg, bg, stride, pad, zin, zout)
    y.backward(pack(gys, zout).to(DEV))
    torch.cuda.synchronize()
    for g in range(G):
        x, w, b, yr = refs[g]
        assert rel(unpack(y, zout, g), yr) < 2e-6, g
        assert rel(unpack(xg.grad, zin, g), x.grad) < 1e-5, g
        assert rel(wg[g].grad, w.grad) < 1e-5, g
        if bias:
            assert rel(bg[g].grad, b.grad) < 1e-5, g
    # ... and bit-for-bit what the per-twin launches of the same kernels produce in the forward pass
    if zin == "stack" and zout == "stack":
        for g in range(G):
            yt = ops.conv3d(cl(xs[g]).to(DEV), ws[g].to(DEV), None, stride, pad)
            assert torch.equal(yt, y.detach()[g * B:(g + 1) * B]), g


@pytest.mark.parametrize("relu_in,relu_out,res,C,training", [(False, True, False, 64, True), (False, True, True, 256, True),
                                                             (True, False, False, 64, True), (False, True, True, 128, False)])
def test_grouped_batch_norm_equals_its_twins(ops, relu_in, relu_out, res, C, training):
    """ops.batch_norm_grouped (three twin nn.BatchNorm3d on batch-stacked activations: per-group statistics, affine parameters and
    running buffers) against F.batch_norm per group on the CPU, forward, input / residual / affine gradients, running statistics."""
    G, B, D, Hh, W = 3, 2, 3, 9, 11
    xs = [rnd(B, C, D, Hh, W, seed=80 + g, scale=1.0 + g) + g for g in range(G)]
    rs = [rnd(B, C, D, Hh, W, seed=90 + g) for g in range(G)]
    gam = [rnd(C, seed=100 + g) * 0.3 + 1 for g in range(G)]
    bet = [rnd(C, seed=110 + g) * 0.1 for g in range(G)]
    rms = [rnd(C, seed=120 + g) * 0.1 for g in range(G)]
    rvs = [rnd(C, seed=130 + g).abs() + 0.5 for g in range(G)]
    gys = [rnd(B, C, D, Hh, W, seed=140 + g) for g in range(G)]
    refs = []
    for g in range(G):
        x, r = xs[g].clone().requires_grad_(), rs[g].clone().requires_grad_()
        ga, be = gam[g].clone().requires_grad_(), bet[g].clone().requires_grad_()
        rm, rv = rms[g].clone(), rvs[g].clone()
        y = F.batch_norm(F.relu(x) if relu_in else x, rm, rv, ga, be, training, 0.1, 1e-5)
        if res:
            y = y + r
        if relu_out:
            y = F.relu(y)
        y.backward(gys[g])
        refs.append((x, r, ga, be, rm, rv, y))
    xg = torch.cat([cl(t) for t in xs], 0).to(DEV).requires_grad_()
    rg = torch.cat([cl(t) for t in rs], 0).to(DEV).requires_grad_() if res else None
    gg = [t.to(DEV).requires_grad_() for t in gam]
    bg = [t.to(DEV).requires_grad_() for t in bet]
    rmg, rvg = [t.to(DEV) for t in rms], [t.to(DEV) for t in rvs]
    y = ops.batch_norm_grouped(xg, gg, bg, rmg, rvg, rg, relu_in, relu_out, training)
    y.backward(torch.cat([cl(t) for t in gys], 0).to(DEV))
    torch.cuda.synchronize()
    for g in range(G):
        x, r, ga, be, rm, rv, yr = refs[g]
        sl = slice(g * B, (g + 1) * B)
        assert rel(ncdhw(y[sl]), yr) < 2e-6
        assert rel(ncdhw(xg.grad[sl]), x.grad) < 2e-5
        if res:
            assert rel(ncdhw(rg.grad[sl]), r.grad) < 1e-6
        assert rel(gg[g].grad, ga.grad) < 2e-5 and rel(bg[g].grad, be.grad) < 2e-5
        assert rel(rmg[g], rm) < 1e-6 and rel(rvg[g], rv) < 1e-6


def test_split_bf16_main_loop_is_at_least_as_accurate_as_the_fp32_mfma_chain(ops, monkeypatch):
    """ops.SPLIT_BF16 (default): the 128-row tiles form every fp32 product from six bf16 MFMA products of exactly split operands
    (csrc/igemm_fwd.h SPLIT).  A long-K convolution on the 128x128 tile (1x3x3, 128 channels, 25088 rows: K = 1152) and a 1x1 with
    K = 2048 against fp64: the split loop must be inside the per-kernel bar and no worse than the v_mfma_f32_32x32x2_f32 chain it replaces
    (measured 0.4-0.8x its error: one fp32 rounding per 16 products of the K sum instead of 16), forward, data gradient and weight gradient."""
    cases = [(128, 128, (1, 3, 3), (0, 1, 1), (2, 4, 56, 56)), (2048, 256, (1, 1, 1), (0, 0, 0), (2, 4, 56, 56))]
    for Ci, Co, k, pad, (B, D, Hh, W) in cases:
        x, w = rnd(B, Ci, D, Hh, W, seed=1), rnd(Co, Ci, *k, seed=2, scale=(Ci * k[1] * k[2]) ** -0.5)
        gy = rnd(B, Co, D, Hh, W, seed=3)
        x64, w64 = x.double().to(DEV).requires_grad_(), w.double().to(DEV).requires_grad_()
        with torch.backends.cudnn.flags(enabled=False):
            truth = F.conv3d(x64, w64, None, 1, pad)
            truth.backward(gy.double().to(DEV))
        errs = {}
        for on in (False, True):
            monkeypatch.setattr(ops, "SPLIT_BF16", on)
            xg, wg = cl(x).to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
            y = ops.conv3d(xg, wg, None, (1, 1, 1), pad)
            y.backward(cl(gy).to(DEV))
            torch.cuda.synchronize()
            errs[on] = (rel(ncdhw(y), truth), rel(ncdhw(xg.grad), x64.grad), rel(wg.grad, w64.grad))
        for e_split, e_f32 in zip(errs[True], errs[False]):
            assert e_split < 2e-6 and e_f32 < 2e-6, errs
            assert e_split <= 1.05 * e_f32, errs


def test_conv3d_sliced_io(ops):
    """input is a channel slice of a wider buffer and the output is written into a slice (in-place concat)."""
    B, D, Hh, W, Ci, Co = 2, 3, 6, 6, 16, 24
    x = rnd(B, Ci, D, Hh, W, seed=5)
    w = rnd(Co, Ci, 1, 1, 1, seed=6, scale=0.25)
    wide = torch.zeros(B, D, Hh, W, 40, device=DEV)
    wide[..., 8:24] = cl(x).to(DEV)
    out = torch.full((B, D, Hh, W, 72), 7.0, device=DEV)
    y = ops.conv3d(wide[..., 8:24], w.to(DEV), None, (1, 1, 1), (0, 0, 0), False, out[..., 24:48])
    torch.cuda.synchronize()
    assert y.data_ptr() == out[..., 24:48].data_ptr()
    assert rel(ncdhw(out[..., 24:48]), F.conv3d(x, w)) < 2e-6
    assert (out[..., :24] == 7).all() and (out[..., 48:] == 7).all()


@pytest.mark.parametrize("shape", [(2, 3, 30, 34), (3, 4, 64, 64), (1, 2, 45, 19)], ids=["ragged", "full_tiles_many", "odd"])
@pytest.mark.parametrize("stem_kernel", [True, False], ids=["patch_kernel", "scalar_gather_gemm"])
def test_stem_conv(ops, shape, stem_kernel, monkeypatch):
    """the encoder stem (Cin = 1, 3x7x7, stride (1,2,2)) on a strided modality plane of the NCDHW input: the patch-staged kernels and the
    generic scalar-gather implicit GEMM they replace, against F.conv3d; ragged tiles, several tiles per persistent workgroup, odd sizes"""
    monkeypatch.setattr(ops, "USE_STEM_KERNEL", stem_kernel)
    B, D, Hh, W = shape
    xin = rnd(B, 3, D, Hh, W, seed=7)
    w = rnd(64, 1, 3, 7, 7, seed=8, scale=1 / 12.0)
    wr = w.clone().requires_grad_()
    yr = F.conv3d(xin[:, 1:2], wr, None, (1, 2, 2), (1, 3, 3))
    gy = rnd(*yr.shape, seed=9)
    yr.backward(gy)
    wg = w.to(DEV).requires_grad_()
    yg = ops.conv3d(xin.to(DEV)[:, 1], wg, None, (1, 2, 2), (1, 3, 3))
    yg.backward(cl(gy).to(DEV))
    torch.cuda.synchronize()
    assert rel(ncdhw(yg), yr) < 2e-6
    assert rel(wg.grad, wr.grad) < 2e-6


def test_linear(ops):
    x = rnd(3, 70, 512, seed=1)
    w = rnd(1536, 512, seed=2, scale=0.05)
    b = rnd(1536, seed=3)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.linear(xr, wr, br)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy)
    xg, wg, bg = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    yg = ops.linear(xg, wg, bg)
    yg.backward(gy.to(DEV))
    torch.cuda.synchronize()
    for a, r in ((yg, yr), (xg.grad, xr.grad), (wg.grad, wr.grad), (bg.grad, br.grad)):
        assert rel(a, r) < 2e-6


@pytest.mark.parametrize("relu_in,relu_out,res,C,training", [(False, True, False, 64, True), (False, True, True, 256, True),
                                                               (True, False, False, 64, True), (False, False, False, 2048, True),
                                                               (False, True, True, 128, False)])
def test_batch_norm(ops, relu_in, relu_out, res, C, training):
    B, D, Hh, W = 2, 3, 7, 9
    x = rnd(B, C, D, Hh, W, seed=1) * 2 + 0.5
    gamma, beta = 0.5 + torch.rand(C, generator=torch.Generator().manual_seed(2)), rnd(C, seed=3, scale=0.2)
    rm, rv = rnd(C, seed=4, scale=0.1), 0.5 + torch.rand(C, generator=torch.Generator().manual_seed(5))
    r = rnd(B, C, D, Hh, W, seed=6) if res else None
    xr, gr, br = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rr = r.clone().requires_grad_() if res else None
    rm_r, rv_r = rm.clone(), rv.clone()
    t = F.relu(xr) if relu_in else xr
    yr = F.batch_norm(t, rm_r, rv_r, gr, br, training, 0.1, 1e-5)
    if res:
        yr = yr + rr
    if relu_out:
        yr = F.relu(yr)
    gy = rnd(*yr.shape, seed=7)
    yr.backward(gy)
    xg, gg, bg = cl(x).to(DEV).requires_grad_(), gamma.to(DEV).requires_grad_(), beta.to(DEV).requires_grad_()
    rg = cl(r).to(DEV).requires_grad_() if res else None
    rm_g, rv_g = rm.to(DEV), rv.to(DEV)
    yg = ops.batch_norm(xg, gg, bg, rm_g, rv_g, rg, relu_in, relu_out, training, 0.1, 1e-5)
    yg.backward(cl(gy).to(DEV))
    torch.cuda.synchronize()
    assert rel(ncdhw(yg), yr) < 2e-6
    assert rel(ncdhw(xg.grad), xr.grad) < 1e-5
    assert rel(gg.grad, gr.grad) < 1e-5 and rel(bg.grad, br.grad) < 1e-5
    if res:
        assert rel(ncdhw(rg.grad), rr.grad) < 1e-6
    assert rel(rm_g, rm_r) < 1e-6 and rel(rv_g, rv_r) < 1e-6          # running statistics (momentum 0.1, unbiased var)


@pytest.mark.parametrize("kind,Ci,Co,k,shape", [("bn", 64, 96, (1, 1, 1), (2, 3, 7, 9)), ("bn", 32, 256, (1, 3, 3), (2, 3, 14, 14)),
                                                  ("in", 48, 48, (1, 1, 1), (3, 4, 8, 8)), ("in", 24, 24, (1, 1, 1), (2, 4, 56, 56))])
def test_conv_with_norm_statistics_from_the_epilogue(ops, kind, Ci, Co, k, shape):
    """conv -> BatchNorm (batch statistics) / conv -> ReLU -> InstanceNorm with the sum / sum-of-squares partials produced by the GEMM
    epilogue (CorrifGemm.stats_part) instead of a separate pass over the conv output"""
    B, D, Hh, W = shape
    pad = (0, k[1] // 2, k[2] // 2)
    x = rnd(B, Ci, D, Hh, W, seed=1)
    w = rnd(Co, Ci, *k, seed=2, scale=1.0 / math.sqrt(Ci * k[1] * k[2]))
    b = rnd(Co, seed=3) if kind == "in" else None
    gamma, beta = 0.5 + torch.rand(Co, generator=torch.Generator().manual_seed(4)), rnd(Co, seed=5, scale=0.2)
    rm, rv = torch.zeros(Co), torch.ones(Co)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    cr = F.conv3d(xr, wr, b, 1, pad)
    yr = F.batch_norm(cr, rm.clone(), rv.clone(), gamma, beta, True, 0.1, 1e-5) if kind == "bn" else F.instance_norm(F.relu(cr), eps=1e-5)
    gy = rnd(*yr.shape, seed=6)
    yr.backward(gy)
    xg, wg = cl(x).to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    st = {"G": 1 if kind == "bn" else B, "relu": kind == "in"}
    cg = ops.conv3d(xg, wg, b.to(DEV) if b is not None else None, (1, 1, 1), pad, False, None, stats=st)
    assert "part" in st                                                   # the epilogue really produced the partials
    if kind == "bn":
        rm_g, rv_g = rm.to(DEV), rv.to(DEV)
        yg = ops.batch_norm(cg, gamma.to(DEV), beta.to(DEV), rm_g, rv_g, None, False, False, True, 0.1, 1e-5, None, st)
    else:
        yg = ops.relu_instnorm(cg, 1e-5, None, st)
    yg.backward(cl(gy).to(DEV))
    torch.cuda.synchronize()
    assert rel(ncdhw(yg), yr) < 3e-6
    assert rel(ncdhw(xg.grad), xr.grad) < 2e-5 and rel(wg.grad, wr.grad) < 2e-5
    if kind == "bn":
        rm_r, rv_r = rm.clone(), rv.clone()
        F.batch_norm(cr.detach(), rm_r, rv_r, gamma, beta, True, 0.1, 1e-5)
        assert rel(rm_g, rm_r) < 1e-5 and rel(rv_g, rv_r) < 1e-5


@pytest.mark.parametrize("Ci,Co,shape,Ds", [(16, 8, (2, 16, 16, 32), 0), (8, 8, (3, 24, 12, 16), 4), (32, 16, (2, 9, 10, 20), 0),
                                            (16, 16, (2, 40, 8, 16), 3), (32, 8, (33, 8, 8, 16), 0), (32, 32, (2, 16, 8, 16), 2)],
                         ids=["d1_c1", "d1_c2_y_with_skip_share", "d2_c1_ragged", "d2_c2_y_ragged_blocks", "many_samples_per_workgroup",
                              "32_channels_add_only"])
def test_patch_conv_epilogue_statistics_and_depth_class_add(ops, Ci, Co, shape, Ds):
    """The decoder's general_conv3d_prenorm layers on the patch kernel (mmvit4.py:41-45,225-235: 3x3x3 replicate-padded conv -> ReLU ->
    InstanceNorm3d): the InstanceNorm statistics come out of the convolution's epilogue (CorrifConv3Patch.stats_part, one partial per
    (sample, workgroup, wave)), and for d*_c2 the compact skip branch's share is broadcast-added by depth class in the same epilogue
    (add_src) before the statistics are taken.  Against conv + add + F.instance_norm(F.relu(.)) on the CPU, forward and backward."""
    B, D, Hh, W = shape
    x = rnd(B, Ci, D, Hh, W, seed=1)
    w = rnd(Co, Ci, 3, 3, 3, seed=2, scale=(27 * Ci) ** -0.5)
    b = rnd(Co, seed=3)
    ys = rnd(B, Co, 3 * Ds, Hh, W, seed=4) if Ds else None
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    cr = F.conv3d(F.pad(xr, (1,) * 6, mode="replicate"), wr, br)
    if Ds:
        src = F.interpolate(torch.arange(Ds, dtype=torch.float32).view(1, 1, Ds, 1, 1), size=(D, 1, 1), mode="nearest").view(-1).long()
        cls = torch.tensor([3 * int(src[d]) + (0 if d == 0 or src[d - 1] != src[d] else 2 if d == D - 1 or src[d + 1] != src[d] else 1)
                            for d in range(D)])
        ysr = ys.clone().requires_grad_()
        cr = cr + ysr[:, :, cls]
    yr = F.instance_norm(F.relu(cr), eps=1e-5)
    gy = rnd(*yr.shape, seed=6)
    yr.backward(gy)
    xg, wg, bg = cl(x).to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    st = {"G": B, "relu": True, "after_bcast": bool(Ds)}
    if Ds:
        ysg = cl(ys).to(DEV).requires_grad_()
        hb = {"ys": ysg, "done": False}
        cg = ops.conv3d(xg, wg, bg, (1, 1, 1), (1, 1, 1), True, None, stats=st, bcast=hb)
        assert hb["done"]                                                 # the epilogue really added the skip share
        cg = ops.depth_bcast_add(cg, ysg, fused=True)
    else:
        cg = ops.conv3d(xg, wg, bg, (1, 1, 1), (1, 1, 1), True, None, stats=st)
    assert ("part" in st) == (Co <= 16)                                   # 8- / 16-channel layers: statistics from the epilogue
    yg = ops.relu_instnorm(cg, 1e-5, None, st)
    yg.backward(cl(gy).to(DEV))
    torch.cuda.synchronize()
    assert rel(ncdhw(yg), yr) < 3e-6
    assert rel(ncdhw(xg.grad), xr.grad) < 2e-5 and rel(wg.grad, wr.grad) < 2e-5 and rel(bg.grad, br.grad) < 2e-5
    if Ds:
        assert rel(ncdhw(ysg.grad), ysr.grad) < 2e-5


@pytest.mark.parametrize("C,shape", [(24, (2, 3, 10, 10)), (192, (3, 8, 8, 8)), (8, (2, 16, 16, 16))])
def test_relu_instnorm(ops, C, shape):
    B, D, Hh, W = shape
    x = rnd(B, C, D, Hh, W, seed=1) + 0.3
    xr = x.clone().requires_grad_()
    yr = F.instance_norm(F.relu(xr), eps=1e-5)
    gy = rnd(*yr.shape, seed=2)
    yr.backward(gy)
    xg = cl(x).to(DEV).requires_grad_()
    out = torch.zeros(B, D, Hh, W, C + 16, device=DEV)
    yg = ops.relu_instnorm(xg, 1e-5, out[..., 8:8 + C])
    yg.backward(cl(gy).to(DEV))
    torch.cuda.synchronize()
    assert rel(ncdhw(out[..., 8:8 + C]), yr) < 2e-6
    assert rel(ncdhw(xg.grad), xr.grad) < 1e-5


def test_layer_norm_with_pos(ops):
    B, N, C = 3, 40, 512
    x, pos = rnd(B, N, C, seed=1), rnd(1, N, C, seed=2, scale=0.1)
    g, b = 0.5 + torch.rand(C, generator=torch.Generator().manual_seed(3)), rnd(C, seed=4, scale=0.1)
    xr, pr, gr, br = [t.clone().requires_grad_() for t in (x, pos, g, b)]
    xs_r = xr + pr
    yr = F.layer_norm(xs_r, (C,), gr, br, 1e-5)
    g1, g2 = rnd(B, N, C, seed=5), rnd(B, N, C, seed=6)
    (yr * g1 + xs_r * g2).sum().backward()
    xg, pg, gg, bg = [t.to(DEV).requires_grad_() for t in (x, pos, g, b)]
    xs, y = ops.layer_norm(xg, gg, bg, pos=pg)
    torch.autograd.backward([y, xs], [g1.to(DEV), g2.to(DEV)])
    torch.cuda.synchronize()
    assert rel(y, yr) < 2e-6 and rel(xs, xs_r) < 1e-7
    assert rel(xg.grad, xr.grad) < 1e-5 and rel(pg.grad, pr.grad) < 1e-5
    assert rel(gg.grad, gr.grad) < 1e-5 and rel(bg.grad, br.grad) < 1e-5
    # plain form
    x2 = x.to(DEV).requires_grad_()
    y2 = ops.layer_norm(x2, gg.detach(), bg.detach())
    assert rel(y2, F.layer_norm(x, (C,), g, b, 1e-5)) < 2e-6


def test_maxpool_with_ties(ops):
    B, C, D, Hh, W = 2, 64, 3, 13, 14
    x = torch.relu(rnd(B, C, D, Hh, W, seed=1)) * 0.5 + 0.25          # many exact ties (the stem is ReLU -> BN -> pool)
    xr = x.clone().requires_grad_()
    yr = F.max_pool3d(xr, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    gy = rnd(*yr.shape, seed=2)
    yr.backward(gy)
    xg = cl(x).to(DEV).requires_grad_()
    yg = ops.maxpool133(xg)
    yg.backward(cl(gy).to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(ncdhw(yg).cpu(), yr.detach())
    assert rel(ncdhw(xg.grad), xr.grad) < 1e-6


@pytest.mark.parametrize("src,dst", [((4, 14, 14), (8, 8, 8)), ((3, 56, 56), (8, 8, 8)), ((8, 8, 8), (16, 16, 16)),
                                      ((12, 12, 12), (1, 21, 21)), ((4, 7, 7), (8, 8, 8)), ((1, 5, 5), (8, 8, 8))])
def test_trilinear(ops, src, dst):
    B, C = 2, 16
    x = rnd(B, C, *src, seed=1)
    xr = x.clone().requires_grad_()
    yr = F.interpolate(xr, size=dst, mode="trilinear", align_corners=True)
    gy = rnd(*yr.shape, seed=2)
    yr.backward(gy)
    xg = cl(x).to(DEV).requires_grad_()
    yg = ops.trilinear(xg, dst)
    yg.backward(cl(gy).to(DEV))
    torch.cuda.synchronize()
    assert rel(ncdhw(yg), yr) < 1e-6
    assert rel(ncdhw(xg.grad), xr.grad) < 2e-6


@pytest.mark.parametrize("src,dst,B,C", [((16, 16, 16), (32, 32, 32), 2, 16), ((8, 8, 8), (16, 16, 16), 3, 8), ((5, 9, 3), (11, 9, 12), 2, 4),
                                          ((1, 4, 4), (6, 9, 8), 1, 12), ((7, 6, 5), (7, 6, 5), 2, 4)])
def test_trilinear_adjoint_one_axis_per_pass(ops, src, dst, B, C):
    """corrif_trilinear_bwd_sep (the decoder's up-samplings: D, H, W passes, every incoming gradient read once) against ATen's
    upsample_trilinear3d_backward and against the one-pass gather, for x2 cubes, ragged ratios, a size-1 source axis and the identity;
    through `ops.trilinear` the first case (>= 2^20 gradient elements) takes the same route (bit-identical result)."""
    import corrif_hip as hip
    x = rnd(B, C, *src, seed=1)
    xr = x.clone().requires_grad_()
    yr = F.interpolate(xr, size=dst, mode="trilinear", align_corners=True)
    gy = rnd(*yr.shape, seed=2)
    yr.backward(gy)
    g = cl(gy).to(DEV)
    need = hip.lib().corrif_trilinear_bwd_sep_workspace(B, C, *src, *dst)
    assert need == 4 * B * C * (src[0] * dst[1] * dst[2] + src[0] * src[1] * dst[2])
    ws = torch.empty(need // 4, device=DEV)
    gx = torch.full((B, *src, C), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert hip.lib().corrif_trilinear_bwd_sep(g.data_ptr(), gx.data_ptr(), ws.data_ptr(), B, C, *src, *dst, st) == 0
    gg = torch.full((B, *src, C), float("nan"), device=DEV)
    assert hip.lib().corrif_trilinear_bwd(g.data_ptr(), C, gg.data_ptr(), C, B, C, *src, *dst, st) == 0
    torch.cuda.synchronize()
    assert rel(ncdhw(gx), xr.grad) < 2e-6
    assert rel(gx, gg) < 1e-6
    # a down-sampling axis is refused (the gather handles it)
    assert hip.lib().corrif_trilinear_bwd_sep_workspace(B, C, 9, 8, 8, 8, 16, 16) == -1
    if gy.numel() >= (1 << 20):
        xg = cl(x).to(DEV).requires_grad_()
        ops.trilinear(xg, dst).backward(g)
        torch.cuda.synchronize()
        assert torch.equal(xg.grad, gx)


def test_trilinear_scale2_equals_upsample(ops):
    x = rnd(1, 8, 16, 16, 16, seed=3)
    yr = torch.nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)(x)
    yg = ops.trilinear(cl(x).to(DEV), (32, 32, 32))
    assert rel(ncdhw(yg), yr) < 1e-6


@pytest.mark.parametrize("src,dst", [((4, 14, 14), (16, 16, 16)), ((3, 28, 28), (32, 32, 32)), ((4, 56, 56), (20, 128, 128))])
def test_nearest(ops, src, dst):
    B, C = 2, 24
    x = rnd(B, C, *src, seed=1)
    xr = x.clone().requires_grad_()
    yr = F.interpolate(xr, dst)
    gy = rnd(*yr.shape, seed=2)
    yr.backward(gy)
    xg = cl(x).to(DEV).requires_grad_()
    yg = ops.nearest(xg, dst)
    yg.backward(cl(gy).to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(ncdhw(yg).cpu(), yr.detach())
    assert rel(ncdhw(xg.grad), xr.grad) < 2e-6


@pytest.mark.parametrize("Ds,D", [(3, 24), (3, 96), (3, 32), (12, 128), (5, 64), (4, 9)],
                         ids=["f8", "f32", "3_to_32_ragged", "12_to_128_ragged", "5_to_64_ragged", "4_to_9_blocks_of_2_3"])
def test_depth_class_broadcast_and_reduce(ops, Ds, D):
    """y[b,d] += ys[b, cls(d)] and its adjoint (class 3k + {0 first, 1 interior, 2 last} of up-sampling block k = the slices whose
    nearest-neighbour source is k), for depth ratios that divide (4 / 8 bands) and that do not (the reference-native 3 bands, 12 bands:
    blocks of floor / ceil(D / Ds) slices); and the identity the decoder relies on: a replicate-padded 3x3x3 conv of a
    depth-nearest-up-sampled tensor equals the broadcast of the same conv on the compact 3-slices-per-block grid.  The block structure
    is taken from F.interpolate itself."""
    B, Hh, W, C = 2, 5, 8, 8
    src = F.interpolate(torch.arange(Ds, dtype=torch.float32).view(1, 1, Ds, 1, 1), size=(D, 1, 1), mode="nearest").view(-1).long()
    cls = torch.tensor([3 * int(src[d]) + (0 if d == 0 or src[d - 1] != src[d] else 2 if d == D - 1 or src[d + 1] != src[d] else 1)
                        for d in range(D)])
    y = rnd(B, D, Hh, W, C, seed=1)
    ys = rnd(B, 3 * Ds, Hh, W, C, seed=2)
    yg = y.clone().to(DEV).requires_grad_()
    sg = ys.clone().to(DEV).requires_grad_()
    out = ops.depth_bcast_add(yg * 1.0, sg)
    go = rnd(B, D, Hh, W, C, seed=3)
    out.backward(go.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(out.detach().cpu(), y + ys[:, cls])
    gs_ref = torch.zeros_like(ys).index_add_(1, cls, go)
    assert torch.equal(yg.grad.cpu(), go) and rel(sg.grad, gs_ref) < 1e-6
    # the convolution identity, on the CPU with stock ops (F.interpolate is what the reference calls at mmvit4.py:271-286)
    x = rnd(1, 4, Ds, 6, 6, seed=4)
    w = rnd(5, 4, 3, 3, 3, seed=5)
    conv = lambda t: F.conv3d(F.pad(t, (1, 1, 1, 1, 1, 1), mode="replicate"), w)
    full = conv(F.interpolate(x, size=(D, 6, 6), mode="nearest"))
    compact = conv(F.interpolate(x, size=(3 * Ds, 6, 6), mode="nearest"))
    assert (full - compact[:, :, cls]).abs().max().item() < 1e-5


@pytest.mark.parametrize("B,N", [(2, 512), (1, 2048)])
@pytest.mark.parametrize("flash", [True, False], ids=["flash", "materialised"])
def test_attention(ops, B, N, flash, monkeypatch):
    monkeypatch.setattr(ops, "FLASH_ATTENTION", flash)
    heads, C = 8, 512
    qkv = rnd(B, N, 3 * C, seed=1, scale=0.5)
    qr = qkv.clone().requires_grad_()
    q, k, v = qr.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    a = torch.softmax((q @ k.transpose(-2, -1)) * (C // heads) ** -0.5, -1)
    outr = (a @ v).transpose(1, 2).reshape(B, N, C)
    go = rnd(B, N, C, seed=2)
    outr.backward(go)
    qg = qkv.to(DEV).requires_grad_()
    og = ops.attention(qg, heads, 0.1, False)
    og.backward(go.to(DEV))
    torch.cuda.synchronize()
    assert rel(og, outr) < 3e-6
    assert rel(qg.grad, qr.grad) < 1e-5


@pytest.mark.parametrize("flash", [True, False], ids=["flash", "materialised"])
def test_attention_dropout_fused_equals_unfused(ops, flash, monkeypatch):
    """the flash kernels / the fused softmax+dropout kernels use the same Philox stream and indexing as the element-wise dropout
    kernel: with one (seed, offset) all three statements of attention draw the identical mask"""
    monkeypatch.setattr(ops, "FLASH_ATTENTION", flash)
    B, N, heads, C = 2, 512, 8, 512
    qkv = rnd(B, N, 3 * C, seed=3, scale=0.5).to(DEV)
    go = rnd(B, N, C, seed=4).to(DEV)
    ops.manual_seed(77)
    q1 = qkv.clone().requires_grad_()
    o1 = ops.attention(q1, heads, 0.1, True)
    o1.backward(go)
    # unfused statement of the same computation with the stand-alone kernels
    ops.manual_seed(77)
    q2 = qkv.clone().requires_grad_()
    qq, kk, vv = q2.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    P_ = torch.softmax((qq @ kk.transpose(-2, -1)) * (C // heads) ** -0.5, -1)
    Pd = ops.dropout(P_.contiguous(), 0.1, True)
    o2 = (Pd @ vv).transpose(1, 2).reshape(B, N, C)
    o2.backward(go)
    torch.cuda.synchronize()
    assert rel(o1, o2) < 3e-6
    assert rel(q1.grad, q2.grad) < 2e-5
    assert (o1 - ops.attention(qkv, heads, 0.1, False)).abs().max().item() > 1e-3     # dropout really is active


def test_flash_attention_matches_materialised_at_2048_tokens(ops, monkeypatch):
    """N = 2048 (the multimodal transformer), dropout on, odd batch: the flash kernels against the materialised-score path with the
    same Philox reservation - outputs and all three gradients - and bit-identical reruns (no atomics)."""
    B, N, heads, C = 3, 2048, 8, 512
    qkv = rnd(B, N, 3 * C, seed=5, scale=0.5).to(DEV)
    go = rnd(B, N, C, seed=6).to(DEV)
    res = {}
    for flash in (True, False, True):
        monkeypatch.setattr(ops, "FLASH_ATTENTION", flash)
        ops.manual_seed(91)
        ops.dropout(torch.ones(64, device=DEV), 0.5, True)          # a non-zero stream offset
        q = qkv.clone().requires_grad_()
        o = ops.attention(q, heads, 0.1, True)
        o.backward(go)
        res.setdefault(flash, []).append((o.detach(), q.grad))
    torch.cuda.synchronize()
    (o1, g1), (o3, g3) = res[True]
    (o2, g2), = res[False]
    assert torch.equal(o1, o3) and torch.equal(g1, g3)
    assert rel(o1, o2) < 3e-6
    for sl in (slice(0, C), slice(C, 2 * C), slice(2 * C, 3 * C)):      # dq, dk, dv
        assert rel(g1[..., sl], g2[..., sl]) < 2e-5


@pytest.mark.parametrize("B", [1, 2, 3, 4, 8, 16, 32, 64])
def test_inter_corr(ops, B):
    """the element-wise cross-modal correlation INCLUDING the batch-dependent re-view (i, b) = divmod(3 b' + i', B) of
    mmvit4.py:481-487, forward and backward, against the CPU oracle - at the small batches of the reference fixtures and at the batches
    BASELINE's configurations run per GPU (8 / 16 / 32 / 64)."""
    from oracle import mmvit4_oracle as O
    S, C = 512, 512
    qkvs = [rnd(B, S, 3 * C, seed=10 + i) for i in range(3)]
    refs = [t.clone().requires_grad_() for t in qkvs]

    def vol(t):       # [B,S,C] tokens -> [B,C,8,8,8]
        return t.reshape(B, 8, 8, 8, C).permute(0, 4, 1, 2, 3)

    q = [vol(t[..., :C]) for t in refs]
    k = [vol(t[..., C:2 * C]) for t in refs]
    v = [vol(t[..., 2 * C:]) for t in refs]
    outs_r = [O.inter_corr(q[m], k, v).permute(0, 2, 3, 4, 1).reshape(B, S, C) for m in range(3)]
    gs = [rnd(B, S, C, seed=20 + i) for i in range(3)]
    torch.autograd.backward(outs_r, gs)
    gin = [t.to(DEV).requires_grad_() for t in qkvs]
    outs = ops.inter_corr(*gin)
    torch.autograd.backward(list(outs), [g.to(DEV) for g in gs])
    torch.cuda.synchronize()
    for m in range(3):
        assert rel(outs[m], outs_r[m]) < 2e-6
        assert rel(gin[m].grad, refs[m].grad) < 1e-5


def test_dropout_stream(ops):
    x = torch.ones(1 << 20, device=DEV).requires_grad_()
    ops.manual_seed(123)
    y = ops.dropout(x, 0.1, True)
    y.backward(torch.ones_like(y))
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.9) < 2e-3                                   # Bernoulli(0.9), n = 1M: sigma = 3e-4
    assert torch.allclose(y[y != 0], torch.tensor(1 / 0.9, device=DEV))
    assert torch.equal(x.grad, y.detach())                          # backward regenerates the same mask
    ops.manual_seed(123)
    assert torch.equal(ops.dropout(x.detach(), 0.1, True), y.detach())
    y2 = ops.dropout(x.detach(), 0.1, True)                         # the stream advances: a different mask
    assert not torch.equal(y2, y.detach())
    assert ops.dropout(x, 0.1, False) is x                          # eval: identity
    # lag-1 autocorrelation of the mask ~ 0
    m = (y.detach() != 0).float() - keep
    assert abs((m[1:] * m[:-1]).mean().item()) < 1e-3


def test_add_gelu_cat(ops):
    a, b = rnd(2, 64, 512, seed=1), rnd(2, 64, 512, seed=2)
    ag = a.to(DEV).requires_grad_()
    y = ops.gelu(ops.add(ag, b.to(DEV)))
    ar = a.clone().requires_grad_()
    yr = F.gelu(ar + b)
    g = rnd(2, 64, 512, seed=3)
    y.backward(g.to(DEV))
    yr.backward(g)
    assert rel(y, yr) < 1e-6 and rel(ag.grad, ar.grad) < 1e-6
    parts = [rnd(2, n, 512, seed=4 + i) for i, n in enumerate((512, 512, 512, 512))]
    pg = [p.to(DEV).requires_grad_() for p in parts]
    c = ops.cat_tokens(*pg)
    assert torch.equal(c.cpu(), torch.cat(parts, 1))
    gg = rnd(2, 2048, 512, seed=9)
    c.backward(gg.to(DEV))
    assert torch.equal(pg[2].grad.cpu(), gg[:, 1024:1536])


def test_head_and_loss(ops):
    B = 2
    x = rnd(B, 1, 224, 224, 8, seed=1)
    w, b = rnd(3, 8, 1, 1, 1, seed=2, scale=0.4), rnd(3, seed=3, scale=0.1)
    _, mask = helpers.make_inputs(B, 3, 8, 8)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    pr = torch.sigmoid(F.conv3d(xr.permute(0, 4, 1, 2, 3), wr, br))
    lr = F.binary_cross_entropy_with_logits(pr, mask)
    lr.backward()
    xg, wg, bg = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    pg = ops.head(xg, wg, bg)
    lg = ops.bce_with_logits_mean(pg, mask.to(DEV))
    lg.backward()
    torch.cuda.synchronize()
    assert pg.shape == (B, 3, 1, 224, 224)
    assert rel(pg, pr) < 1e-6 and abs(lg.item() - lr.item()) < 1e-6
    assert rel(xg.grad, xr.grad) < 1e-5 and rel(wg.grad, wr.grad) < 1e-5 and rel(bg.grad, br.grad) < 1e-5


def test_jaccard_matches_reference_fixture(ops):
    """F5_JACCARD2.py through the device kernel: bit-identical on 0/1 masks, 1e-6 on soft predictions."""
    g = np.load(os.path.join(helpers.GOLDEN, "jaccard.npz"))
    gen = torch.Generator().manual_seed(7)
    n = 2 * 224 * 224
    y = (torch.rand(n, 1, generator=gen) > 0.6).float()
    soft = torch.rand(n, 1, generator=gen)
    hard = (soft > 0.5).float()
    zero = torch.zeros(n, 1)
    cases = {"soft": (y, soft), "hard": (y, hard), "allzero_mask": (zero, soft), "allzero_hard": (zero, hard), "perfect": (y, y.clone())}
    for nm, (a, b) in cases.items():
        out = ops.jaccard_all(a.to(DEV), b.to(DEV)).cpu().numpy()
        exact = nm in ("hard", "allzero_hard", "perfect")
        for i, key in enumerate(("j2_", "j1_", "f1_")):
            ref = g[key + nm]
            if exact:
                assert out[i] == ref[0], (nm, key, out[i], ref)        # integer partial sums: bit-identical
            else:
                assert abs(out[i] - ref[0]) <= 1e-6 * max(1.0, abs(ref[0])), (nm, key)


def test_adam_matches_torch(ops):
    p0, g0 = rnd(1000, seed=1), rnd(1000, seed=2)
    pr = p0.clone().requires_grad_()
    opt = torch.optim.Adam([pr], lr=1e-3)
    p = p0.to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        pr.grad = g0 * step
        opt.step()
        helpers_status = ops.lib().corrif_adam_step(p.data_ptr(), (g0 * step).to(DEV).data_ptr(), m.data_ptr(), v.data_ptr(), 1000, 1e-3, 0.9,
                                                    0.999, 1e-8, 0.0, step, ops.stream())
        assert helpers_status == 0
    torch.cuda.synchronize()
    assert rel(p, pr) < 1e-6
