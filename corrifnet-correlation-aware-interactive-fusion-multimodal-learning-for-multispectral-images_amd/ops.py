"""autograd.Function wrappers over the C-ABI kernels (corrif_hip.py).

Every forward/backward below enqueues hand-written gfx950 kernels only; torch supplies memory, views and the autograd graph.
What autograd itself still launches on the training step (stock ATen, ~25 small launches, 0.3 ms): the accumulation of the second
sample-group lane's gradient into the decoder / multimodal-transformer parameters and of five two-consumer token tensors.  Activations are channels-last: [B, D, H, W, C] (or [B, N, C] tokens),
possibly as a channel-slice view of a wider concat buffer (row pitch `ld` > C).
"""
import math
import os

import torch
from torch.autograd import Function

import corrif_hip as H
from corrif_hip import P, check, lib, stream

NORM_RELU_IN, NORM_RELU_OUT = 1, 2
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2

# ---- TORCH_LIBRARY binding of the same C-ABI (csrc_torch/corrif_torch.cpp -> libcorrif_torch.so): torch.ops.corrif.* for the NO-GRAD
# forward of the convolution / BatchNorm / InstanceNorm / Linear families - the op validates, allocates and fills the C-ABI descriptor
# in C++, so an eval forward (F4_TRAIN.py:181-208, allJaccardResults_irem_f1_jcrd.py:201-222: launch-bound at batch 1) spends no Python
# per launch on these ~430 of its ~700 launches.  The training path (autograd.Function objects below) uses the ctypes binding.
# CORRIF_TORCH_LIBRARY=0 (or ops.USE_TORCH_LIBRARY = False) routes everything through ctypes (A/B, diagnostics).
USE_TORCH_LIBRARY = os.environ.get("CORRIF_TORCH_LIBRARY", "1") != "0"
_tl = None


def tl():
    """torch.ops.corrif, loading libcorrif_torch.so on first use; None when the binding is switched off"""
    global _tl
    if not USE_TORCH_LIBRARY:
        return None
    if _tl is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcorrif_torch.so")
        if not os.path.exists(path):
            raise RuntimeError("corrif: %s not found - build it with `python __graft_entry__.py` (or set CORRIF_TORCH_LIBRARY=0 to use the "
                               "ctypes binding of the same kernels only)" % path)
        lib()                                   # libcorrif_gfx950.so first: the shim links against it
        torch.ops.load_library(path)
        _tl = torch.ops.corrif
    return _tl


def _fast():
    """the TORCH_LIBRARY ops serve this call: no autograd graph is being recorded and the kernel-variant switches are at their defaults"""
    return USE_TORCH_LIBRARY and not torch.is_grad_enabled() and not (STREAM_K or STREAM_K_LONG) and SPLIT_BF16 and USE_PATCH and \
        USE_STEM_KERNEL and PATCH_STATS


# --------------------------------------------------------------------------------------- helpers
def rows_view(t):
    """Return (t', rows, ld): last dim contiguous, every leading dim dense over a uniform row pitch `ld`,
    16-byte aligned.  Copies (t.contiguous()) only when the layout cannot be described that way."""
    C = t.shape[-1]
    ok = t.stride(-1) == 1 or C == 1
    ld = None
    if ok:
        expect = None
        for i in range(t.dim() - 2, -1, -1):
            if t.shape[i] == 1:
                continue
            if expect is None:
                ld = t.stride(i)
                expect = ld * t.shape[i]
            else:
                if t.stride(i) != expect:
                    ok = False
                    break
                expect *= t.shape[i]
        if ld is None:
            ld = C
        if ld < C or (ld & 3) or (t.data_ptr() & 15):
            ok = False
    if not ok:
        t = t.contiguous()
        ld = C
    return t, t.numel() // C, ld


def empty_like_rows(shape, ref):
    return torch.empty(shape, dtype=torch.float32, device=ref.device)


def _ws(nbytes, dev):
    return H.ws_bytes(nbytes, dev)


# Stream-K split of gemm_fwd launches whose tile grid fills the chip unevenly (corrif_gemm_fwd_workspace / CorrifGemm.ws).  Off by
# default: measured on the B=32 step it is a wash in kernel time (rocprofv3, serial schedule: 118.6 vs 124.4 ms for the family) and
# a loss in wall time (304.9 vs 300.7 ms per step): the branch streams already fill the tails, short-K shapes pay more for the slab
# round trip than they gain, and every launch costs a second descriptor call on the host.
STREAM_K = False
# STREAM_K_LONG restricts it to where it helps a launch in isolation: long K loops on a small tile grid (e4 / e5's 3x3 convolutions and
# their data gradients: 392 / 196 tiles of 128 x 128 on 512 slots, 72 / 144 K tiles; 84 -> 98 TFLOP/s alone, tools/gemm_lab.hip).
STREAM_K_LONG = False     # measured (round 2): wall-neutral on the B=32 step (248.4 vs 248.0 ms), the branch streams already fill those tails
# Grouped launches (Z = 3 twin layers of the modality encoders) with a long K loop: the split pays in isolation (tools/group_microbench.py,
# round 3: e4 conv2 99 -> 106 TFLOP/s, e5 conv2 85 -> 108, e5 conv1 92 -> 102) while short-K shapes lose 8-9 % to the slab round trip
# (K <= 1024); on the B=32 step it is worth 0.8 ms of 254 (A/B, one box) - and it re-associates the K sum, so the grouped schedule would
# no longer be bit-identical to the per-modality one.  Off: the two encoder schedules produce the same bits (tests/test_model_gpu.py).
STREAM_K_GROUPED = False
# Main loop of the GEMM family's 128-row tiles (CorrifGemm.f32_mfma / CorrifWgrad.f32_mfma = 0): fp32 operands split exactly into three
# bf16 terms at the LDS store, six bf16 MFMA products per fp32 product, fp32 accumulation (csrc/igemm_fwd.h "SPLIT").  Against fp64 the
# result is MORE accurate than the fp32-input MFMA chain it replaces (one rounding per 16 products of the K sum instead of 16: 0.36x the
# error at K = 2304-4608, tools/split_lab.hip) and 1.4-1.5x faster.  False = v_mfma_f32_32x32x2_f32 everywhere (rounds 1-3; A/B).
SPLIT_BF16 = True
TRACK_SPLIT = False       # measurement (bench.py): LAST_SPLIT = did the last gemm / wgrad launch take the split-bf16 loop (corrif_*_is_split)
LAST_SPLIT = 0


def gemm(A, lda, Bm, ldb, b_layout, Cout, ldc, M, N, K, Cs, geom, bias=None, addend=None, ld_add=0, act=ACT_NONE,
         Z=1, Zi=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), taps=None, out_map=None, stats=None, addend2=None, ld_add2=0, bstats=None, zs=None):
    g = H.Gemm()
    if zs:                      # grouped launch: per-group strides of the epilogue operands (CorrifGemm.zs_*)
        for k, v in zs.items():
            setattr(g, "zs_" + k, int(v))
    if stats is not None:
        g.stats_part, g.stats_rows_per_group, g.stats_relu = stats
    if taps is not None:
        g.ntap_sel = len(taps)
        for i, t in enumerate(taps):
            g.tap_sel[i] = t
    if out_map is not None:
        g.out_map = 1
        (g.OD, g.OH, g.OW), (g.om_d, g.om_h, g.om_w), (g.oo_d, g.oo_h, g.oo_w) = out_map
    g.A, g.lda, g.Cs = A, lda, Cs
    g.B, g.ldb, g.b_layout = Bm, ldb, b_layout
    g.C, g.ldc = Cout, ldc
    g.bias = bias if bias is not None else None
    g.addend = addend if addend is not None else None
    g.ld_add = ld_add
    g.addend2 = addend2 if addend2 is not None else None
    g.ld_add2 = ld_add2
    if bstats is not None:
        g.bstats_x, g.bstats_ldx, g.bstats_y, g.bstats_ldy, g.bstats_mean, g.bstats_rstd = bstats
    g.M, g.N, g.K, g.act = M, N, K, act
    g.Z, g.Zi = Z, Zi
    g.sA_o, g.sA_i = sA
    g.sB_o, g.sB_i = sB
    g.sC_o, g.sC_i = sC
    g.g = geom
    buf = None
    sk = STREAM_K or (STREAM_K_LONG and K >= 2048 and Z == 1 and M * N <= 128 * 128 * 512) or (STREAM_K_GROUPED and zs is not None and K >= 2048)
    g.no_split = 0 if sk else 1
    g.f32_mfma = 0 if SPLIT_BF16 else 1
    if TRACK_SPLIT:
        global LAST_SPLIT
        LAST_SPLIT = lib().corrif_gemm_fwd_is_split(g)
    if sk:
        nws = lib().corrif_gemm_fwd_workspace(g)      # stream-K split: slabs for the tiles a share boundary cuts
        if nws:
            buf = _ws(nws, torch.device("cuda", torch.cuda.current_device()))
            g.ws = buf.data_ptr()
    check(lib().corrif_gemm_fwd(g, stream()), "corrif_gemm_fwd")


WGRAD_SPLITS_R1 = False      # diagnostics: the round-1 split rule ceil(2048 / tiles) (shorter fp32 accumulation chains, 1.6 waves of workgroups)


def _old_wgrad_splits(R, M, N):
    BM = 32 if M <= 32 else 64
    BN = 256 if (M <= 16 and M % 4 == 0) else 128
    if M >= 128 and N <= 64:
        BM, BN = 128, 64
    if BN == 256:
        BM = 16
    tiles = -(-M // BM) * -(-N // BN)
    return int(max(1, min(-(-2048 // tiles), R // 256, 4096)))


def wgrad(A, lda, Bm, ldb, Cs, Cout, ldc, R, M, N, geom, dev, Z=1, Zi=1, sA=(0, 0), sB=(0, 0), sC=(0, 0)):
    w = H.Wgrad()
    w.A, w.lda = A, lda
    w.B, w.ldb, w.Cs = Bm, ldb, Cs
    w.C, w.ldc = Cout, ldc
    w.R, w.M, w.N = R, M, N
    w.Z, w.Zi = Z, Zi
    w.sA_o, w.sA_i = sA
    w.sB_o, w.sB_i = sB
    w.sC_o, w.sC_i = sC
    w.g = geom
    w.f32_mfma = 0 if SPLIT_BF16 else 1
    if TRACK_SPLIT:
        global LAST_SPLIT
        LAST_SPLIT = lib().corrif_wgrad_is_split(w)
    if Z > 1:                   # grouped weight gradient (Zi = 1): row splits per group; batched attention products (Zi > 1): none
        w.splits = lib().corrif_wgrad_plan(R, M, N, Z) if Zi == 1 else 1
    else:
        w.splits = _old_wgrad_splits(R, M, N) if WGRAD_SPLITS_R1 else lib().corrif_wgrad_plan(R, M, N, 1)
    buf = None
    if w.splits > 1:
        buf = _ws(lib().corrif_wgrad_workspace(w), dev)
        w.ws = buf.data_ptr()
    check(lib().corrif_wgrad(w, stream()), "corrif_wgrad")


def col_sum(t, rows, ld, C):
    out = torch.empty(C, dtype=torch.float32, device=t.device)
    ws = _ws(lib().corrif_col_sum_workspace(rows, C), t.device)
    check(lib().corrif_col_sum(P(t), ld, rows, C, P(out), P(ws), stream()), "corrif_col_sum")
    return out


def repack(src, shape_out, O, I, T, mode, ldo, zero=False):
    out = torch.empty(shape_out, dtype=torch.float32, device=src.device)
    if zero:
        check(lib().corrif_fill(P(out), out.numel(), 0.0, stream()), "corrif_fill")
    check(lib().corrif_weight_repack(P(src), P(out), O, I, T, mode, ldo, stream()), "corrif_weight_repack")
    return out


def conv3_patch(x, ldx, wp, y, ldy, bias, B, S, O, Ci, Co, pad, clamp, cc, fold=False, stats=None, add=None):
    """stats: (partials pointer, chunks, relu) - InstanceNorm statistics of the output from the epilogue; add: (pointer, row pitch, Ds) -
    the compact skip branch's share broadcast-added by depth class in the epilogue (CorrifConv3Patch.stats_part / add_src)"""
    q = H.Conv3Patch()
    q.fold = 1 if fold else 0
    if stats is not None:
        q.stats_part, q.stats_chunks, q.stats_relu = stats
    if add is not None:
        q.add_src, q.ld_add, q.add_Ds = add
    q.X, q.ldx, q.Wp, q.Y, q.ldy = x, ldx, wp, y, ldy
    q.bias = bias
    q.B = B
    q.Sd, q.Sh, q.Sw = S
    q.Od, q.Oh, q.Ow = O
    q.Ci, q.Co, q.pad, q.clamp, q.cc = Ci, Co, pad, 1 if clamp else 0, cc
    check(lib().corrif_conv3_patch(q, stream()), "corrif_conv3_patch")


def conv3_patch_wgrad(x, ldx, gy, ldg, gwp, B, S, O, Ci, Co, clamp, dev):
    q = H.Conv3PatchWgrad()
    ws = _ws(lib().corrif_conv3_patch_wgrad_workspace(Ci, Co), dev)
    q.X, q.ldx, q.DY, q.lddy, q.dW, q.ws = x, ldx, gy, ldg, gwp, ws.data_ptr()
    q.B = B
    q.Sd, q.Sh, q.Sw = S
    q.Od, q.Oh, q.Ow = O
    q.Ci, q.Co, q.pad, q.clamp = Ci, Co, 1, 1 if clamp else 0
    check(lib().corrif_conv3_patch_wgrad(q, stream()), "corrif_conv3_patch_wgrad")


USE_STEM_KERNEL = True      # diagnostics: False routes the encoder stem (Cin = 1, 3x7x7 / (1,2,2)) through the scalar-gather implicit GEMM


def stem_fwd(x, batch_pitch, wp, y, ldy, B, D, Hh, W):
    check(lib().corrif_stem_fwd(P(x), batch_pitch, P(wp), P(y), ldy, B, D, Hh, W, stream()), "corrif_stem_fwd")


def stem_wgrad(x, batch_pitch, gy, ldg, gwp, B, D, Hh, W):
    ws = _ws(lib().corrif_stem_wgrad_workspace(), gy.device)
    check(lib().corrif_stem_wgrad(P(x), batch_pitch, P(gy), ldg, P(gwp), P(ws), B, D, Hh, W, stream()), "corrif_stem_wgrad")


USE_PATCH = True      # diagnostics: False routes the narrow 3x3x3 layers through the implicit-GEMM kernels instead of the patch kernels
PATCH_STATS = True    # diagnostics / A-B: False = the InstanceNorm after a patch-kernel convolution takes its statistics in its own pass
TRILINEAR_SEPARABLE = True   # A-B: False = the adjoint of every trilinear up-sampling is the one-pass gather (rounds 1-2)


def _patch_cc(k, stride, pad, Ci, Co):
    """channel chunk of the patch-staged 3x3x3 kernel for this layer, 0 = use the implicit GEMM"""
    if not USE_PATCH or k != (3, 3, 3) or stride != (1, 1, 1) or pad != (1, 1, 1):
        return 0
    return lib().corrif_conv3_patch_cc(Ci, Co)


def _dgrad_parity_classes(gy, ldg, wd, gx, B, Sin, Sout, k, stride, pad, Ci, Co, G=1, zG=0, zX=0, ldx=None):
    """Data gradient of a strided convolution without multiplying zeros.  Input voxel i receives tap t only if (i + pad - t) is a
    multiple of the stride, so the input grid splits into stride^3 parity classes, each with its own small tap subset
    (3x3 / stride 2: 1, 2, 2 and 4 taps; 1x1 / stride 2: one class with one tap, three classes with none = zeros).  One GEMM per
    class over that class's sub-grid, K = |taps| * Co, rows scattered back with the kernel's output row map.
    wd: [T*Co, Ci] data-gradient weights (tap-major).  gx: [B, Di, Hi, Wi, Ci] is fully written.
    G > 1: grouped launch - wd is [G, T*Co, Ci], gy / gx move by zG / zX floats per group (B = samples per group)."""
    kd, kh, kw = k
    covered = True
    launches = []
    for pd_ in range(stride[0]):
        for ph_ in range(stride[1]):
            for pw_ in range(stride[2]):
                par = (pd_, ph_, pw_)
                R = tuple((Sin[a] - par[a] + stride[a] - 1) // stride[a] for a in range(3))
                if min(R) <= 0:
                    continue
                taps = [(td * kh + th) * kw + tw for td in range(kd) for th in range(kh) for tw in range(kw)
                        if (par[0] + pad[0] - td) % stride[0] == 0 and (par[1] + pad[1] - th) % stride[1] == 0
                        and (par[2] + pad[2] - tw) % stride[2] == 0]
                if not taps:
                    covered = False
                    continue
                launches.append((par, R, taps))
    if not covered:                                 # classes no tap can reach (1x1 stride-2: 3 of 4) keep exact zeros
        check(lib().corrif_fill(P(gx), gx.numel(), 0.0, stream()), "corrif_fill")
    T = kd * kh * kw
    for par, R, taps in launches:
        wc = torch.empty((G, len(taps) * Co, Ci), dtype=torch.float32, device=gx.device)
        for gi in range(G):
            for j, t in enumerate(taps):            # gather this class's tap blocks [Co][Ci] of the tap-major weight matrix
                check(lib().corrif_copy2d(wd.data_ptr() + 4 * (gi * T + t) * Co * Ci, Ci, wc.data_ptr() + 4 * (gi * len(taps) + j) * Co * Ci, Ci,
                                          Co, Ci, 0, stream()), "corrif_copy2d")
        g = H.Geom()
        g.is_gemm = 0
        g.Rd, g.Rh, g.Rw = R
        g.Sd, g.Sh, g.Sw = Sout
        g.kd, g.kh, g.kw = k
        g.mul_d, g.mul_h, g.mul_w = stride
        g.off_d, g.off_h, g.off_w = (par[0] + pad[0], par[1] + pad[1], par[2] + pad[2])
        g.div_d, g.div_h, g.div_w = stride
        g.dir, g.clamp, g.ntaps = -1, 0, kd * kh * kw
        gemm(P(gy), ldg, P(wc), Ci, 1, P(gx), Ci if ldx is None else ldx, B * R[0] * R[1] * R[2], Ci, len(taps) * Co, Co, g, taps=taps,
             out_map=(Sin, stride, par), Z=G, sA=(zG, 0), sB=(len(taps) * Co * Ci, 0), sC=(zX, 0))


def _out_size(i, k, s, p):
    return (i + 2 * p - k) // s + 1


# --------------------------------------------------------------------------------------- convolution
BWD_STATS = True      # diagnostics / A-B: False keeps the separate reduction pass of every BatchNorm backward


def _bwd_stats_request(bl, grad_link, rows, C, dev, xin, ldin):
    """Backward statistics of the BatchNorm that produced this convolution's input, out of the data-gradient GEMM's epilogue
    (CorrifGemm.bstats_*): possible when this GEMM's stored value IS the complete gradient of that BatchNorm's output - the input has no
    other consumer, or all of them parked their gradients on `grad_link` and the epilogue absorbed every one.  Returns (stats, bstats)
    for ops.gemm and leaves the partial buffer in the BatchNorm's link."""
    if not BWD_STATS or bl is None or "x" not in bl or bl["rows"] != rows or bl["C"] != C or C <= 16:
        return None, None
    if grad_link is not None and (grad_link.get("gs") or grad_link.get("late")):
        return None, None                     # a parked gradient was left for an add pass, or a consumer already fell back to autograd
    chunks = (rows + 63) // 64
    part = torch.empty(C * chunks * 2, dtype=torch.float64, device=dev)
    bl["part"], bl["chunks"], bl["grad_link"] = part, chunks, grad_link
    # the ReLU mask of that BatchNorm is its output = this convolution's own input (kept out of the link: no reference cycle)
    return (part.data_ptr(), rows, 0), (P(bl["x"]), bl["ldx"], P(xin) if bl["relu_out"] else None, ldin, P(bl["mean"]), P(bl["rstd"]))


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


# ---- weight gradients on a side stream -------------------------------------------------------------------------------------------
# A convolution's weight gradient is off the backward's critical path (nothing downstream reads it before the optimiser), while the
# BatchNorm backward passes that follow it on the same branch stream are HBM-bound: with the weight-gradient GEMM on a side stream of the
# branch, the matrix pipe stays busy under those passes (round 2 timeline: 30 ms of the step had only normalisation kernels resident).
# Only for parameters that receive exactly one gradient per step and whose .grad is None (autograd then just adopts the tensor - no
# accumulation kernel could race with the side stream); every side stream is joined into the default stream by an autograd-engine
# callback at the end of the backward pass.  Measured (round 2, one box, alternating runs): 251.1 ms per step with it, 250.7 without -
# the normalisation passes and the weight gradients share the CUs either way - so it is OFF; kept as a switch.
SIDE_WGRAD = False
SIDE_WGRAD_GROUPED = True      # the grouped (stacked-modality) encoder: see GroupedConvFn.backward
_side_streams = {}            # id of the stream a backward node runs on -> (side stream, event pool, cursor)
_side_pending = {"queued": False}


def _side_join():
    _side_pending["queued"] = False
    cur = torch.cuda.current_stream()
    for st, _, _ in _side_streams.values():
        cur.wait_stream(st)


def _side_begin(tensors):
    """side stream of the current stream, ordered after everything enqueued on the current stream so far; `tensors` are read there"""
    main = torch.cuda.current_stream()
    ent = _side_streams.get(main.cuda_stream)
    if ent is None:
        ent = _side_streams[main.cuda_stream] = [torch.cuda.Stream(), [], 0]
    side, pool, i = ent
    if i == len(pool):
        pool.append(torch.cuda.Event())
    ev = pool[i]
    ent[2] = (i + 1) % 64                     # ring of persistent events (an event is re-recorded long after its wait was enqueued)
    ev.record(main)
    side.wait_event(ev)
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    if not _side_pending["queued"]:
        _side_pending["queued"] = True
        torch.autograd.Variable._execution_engine.queue_callback(_side_join)
    return side


class ConvFn(Function):
    """nn.Conv3d on channels-last activations.  weight stays in the reference (O,I,kd,kh,kw) layout."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, replicate, out, act, stats_req, grad_link=None, side_ok=False, bwd_stats=None, bcast=None):
        ctx.grad_link = grad_link
        ctx.side_ok = side_ok
        ctx.bwd_stats = bwd_stats
        Co, Ci, kd, kh, kw = weight.shape
        T = kd * kh * kw
        stem = Ci == 1
        if stem:                                  # x: [B, D, H, W] strided view of the NCDHW input, one modality
            B, Di, Hi, Wi = x.shape
            if x.stride(3) != 1 or x.stride(2) != Wi or x.stride(1) != Hi * Wi:
                x = x.contiguous()
            lda, batch_pitch = 1, x.stride(0)
        else:
            x, _, lda = rows_view(x)
            B, Di, Hi, Wi, _ = x.shape
            batch_pitch = 0
        Do, Ho, Wo = (_out_size(Di, kd, stride[0], pad[0]), _out_size(Hi, kh, stride[1], pad[1]), _out_size(Wi, kw, stride[2], pad[2]))
        M = B * Do * Ho * Wo
        if out is None:
            out = torch.empty((B, Do, Ho, Wo, Co), dtype=torch.float32, device=x.device)
        y, _, ldc = rows_view(out)
        assert y is out, "conv output slice must be row-addressable"
        is_gemm = T == 1 and stride == (1, 1, 1) and not stem
        stem_k = stem and bias is None and act == ACT_NONE and not replicate and USE_STEM_KERNEL and \
            lib().corrif_stem_supported(Co, kd, kh, kw, *stride, *pad)
        if stem_k:                                # the encoder stem: patch-staged kernel for Cin = 1, 3x7x7 / (1,2,2)
            wp = repack(weight, (Co, 148), Co, 1, T, 0, 148, zero=True)
            stem_fwd(x, batch_pitch, wp, y, ldc, B, Di, Hi, Wi)
        elif stem:
            Kp = (T + 3) // 4 * 4
            wp = repack(weight, (Co, Kp), Co, 1, T, 0, Kp, zero=True)
            geom = H.conv_geom((Do, Ho, Wo), (Di, Hi, Wi), (kd, kh, kw), stride, pad, clamp=replicate, ntaps=T, src_batch_pitch=batch_pitch)
            gemm(P(x), 1, P(wp), Kp, 0, P(y), ldc, M, Co, Kp, 1, geom, bias=P(bias) if bias is not None else None, act=act)
        elif is_gemm and act == ACT_NONE and lib().corrif_conv1x1_small_supported(Ci, Co):
            check(lib().corrif_conv1x1_small_fwd(P(x), lda, P(weight), 0, P(bias), P(y), ldc, M, Ci, Co, stream()), "corrif_conv1x1_small_fwd")
        elif _patch_cc((kd, kh, kw), stride, pad, Ci, Co) and act == ACT_NONE:
            cc = _patch_cc((kd, kh, kw), stride, pad, Ci, Co)
            wp = repack(weight, (Ci // cc, Co, T, cc), Co, Ci, T, 3, cc)
            add = pst = None
            if bcast is not None:                     # the compact skip branch's share of d*_c2, added by depth class in the epilogue
                bv, _, ldb = rows_view(bcast["ys"])
                assert bv is bcast["ys"] and tuple(bv.shape) == (B, bv.shape[1], Ho, Wo, Co) and bv.shape[1] % 3 == 0
                add = (P(bv), ldb, bv.shape[1] // 3)
                bcast["done"] = True
            if PATCH_STATS and stats_req is not None and stats_req["G"] == B and stats_req["relu"] and \
                    lib().corrif_conv3_patch_stats_supported(Ci, Co):
                chunks = lib().corrif_conv3_patch_stats_chunks(B, Do, Ho, Wo)      # InstanceNorm statistics partials from the epilogue
                part = torch.empty(B * Co * chunks * 2, dtype=torch.float64, device=x.device)
                check(lib().corrif_fill(P(part), 2 * part.numel(), 0.0, stream()), "corrif_fill")
                pst = (part.data_ptr(), chunks, 1)
                stats_req["part"], stats_req["chunks"], stats_req["rpg"] = part, chunks, Do * Ho * Wo
            conv3_patch(P(x), lda, P(wp), P(y), ldc, P(bias), B, (Di, Hi, Wi), (Do, Ho, Wo), Ci, Co, 1, replicate, cc, stats=pst, add=add)
        else:
            wp = weight if T == 1 else repack(weight, (Co, T * Ci), Co, Ci, T, 0, T * Ci)
            geom = H.gemm_geom() if is_gemm else H.conv_geom((Do, Ho, Wo), (Di, Hi, Wi), (kd, kh, kw), stride, pad, clamp=replicate)
            st = None
            if stats_req is not None and Co > 16 and act == ACT_NONE:
                G = stats_req["G"]
                rpg = M // G
                if M % G == 0 and (G == 1 or rpg % 64 == 0):
                    chunks = (rpg + 63) // 64          # sum / sum-of-squares partials for the norm that follows, from the epilogue
                    part = torch.empty(G * Co * chunks * 2, dtype=torch.float64, device=x.device)
                    st = (part.data_ptr(), rpg, 1 if stats_req["relu"] else 0)
                    stats_req["part"], stats_req["chunks"], stats_req["rpg"] = part, chunks, rpg
            gemm(P(x), lda, P(wp), T * Ci, 0, P(y), ldc, M, Co, T * Ci, Ci, geom, bias=P(bias) if bias is not None else None, act=act,
                 stats=st)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, replicate, bias is not None, (B, Di, Hi, Wi), (Do, Ho, Wo), stem, is_gemm, lda, batch_pitch)
        ctx.stem_k = bool(stem_k)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, pad, replicate, has_bias, (B, Di, Hi, Wi), (Do, Ho, Wo), stem, is_gemm, lda, batch_pitch = ctx.cfg
        Co, Ci, kd, kh, kw = weight.shape
        T = kd * kh * kw
        gy, M, ldg = rows_view(gy)
        dev = gy.device
        gx = gw = gb = None
        if ctx.needs_input_grad[0] and not stem:
            gx = torch.empty((B, Di, Hi, Wi, Ci), dtype=torch.float32, device=dev)
            Min = B * Di * Hi * Wi
            if is_gemm and lib().corrif_conv1x1_small_supported(Ci, Co):
                check(lib().corrif_conv1x1_small_fwd(P(gy), ldg, P(weight), 1, 0, P(gx), Ci, Min, Co, Ci, stream()), "corrif_conv1x1_small_fwd")
            elif is_gemm:      # dX[M,Ci] = dY[M,Co] . W[Co,Ci]   (W is the [K][N] operand as stored)
                # gradients of the same tensor parked by its other consumers (grad_tap: the residual branch of an identity block; the
                # downsample and adapt convolutions of a layer output): the GEMM epilogue adds up to two of them, instead of separate
                # accumulation passes over all of them
                adds = []
                link = ctx.grad_link
                while link is not None and link.get("gs") and len(adds) < 2:
                    ga, _, ld_a = rows_view(link["gs"].pop())
                    adds.append((ga, ld_a))
                    TAP_STATS["epilogue"] += 1
                adds += [(None, 0)] * (2 - len(adds))
                st, bst = _bwd_stats_request(ctx.bwd_stats, link, Min, Ci, dev, x, lda)
                gemm(P(gy), ldg, P(weight), Ci, 1, P(gx), Ci, Min, Ci, Co, Co, H.gemm_geom(), addend=P(adds[0][0]) if adds[0][0] is not None else None,
                     ld_add=adds[0][1], addend2=P(adds[1][0]) if adds[1][0] is not None else None, ld_add2=adds[1][1], stats=st, bstats=bst)
            elif _patch_cc((kd, kh, kw), stride, pad, Co, Ci):      # data gradient = patch conv of dY with flipped weights
                cc = _patch_cc((kd, kh, kw), stride, pad, Co, Ci)
                wd = repack(weight, (Co // cc, Ci, T, cc), Co, Ci, T, 4, cc)
                if replicate:
                    Rg = (Di + 2, Hi + 2, Wi + 2)
                    if Di % 4 == 0 and Hi % 4 == 0 and Wi % 16 == 0:      # padding adjoint fused into the kernel's epilogue
                        conv3_patch(P(gy), ldg, P(wd), P(gx), Ci, 0, B, (Do, Ho, Wo), Rg, Co, Ci, 2, False, cc, fold=True)
                    else:
                        gxp = torch.empty((B,) + Rg + (Ci,), dtype=torch.float32, device=dev)
                        conv3_patch(P(gy), ldg, P(wd), P(gxp), Ci, 0, B, (Do, Ho, Wo), Rg, Co, Ci, 2, False, cc)
                        check(lib().corrif_pad_fold(P(gxp), P(gx), Ci, B, Di, Hi, Wi, Ci, stream()), "corrif_pad_fold")
                else:
                    conv3_patch(P(gy), ldg, P(wd), P(gx), Ci, 0, B, (Do, Ho, Wo), (Di, Hi, Wi), Co, Ci, 1, False, cc)
            else:
                wd = weight if T == 1 else repack(weight, (T * Co, Ci), Co, Ci, T, 1, T * Ci)
                if replicate:  # gradient on the replicate-padded grid, then fold the halo back (adjoint of the clamp)
                    assert pad == (1, 1, 1)
                    Rg = (Di + 2, Hi + 2, Wi + 2)
                    gxp = torch.empty((B,) + Rg + (Ci,), dtype=torch.float32, device=dev)
                    if stride == (1, 1, 1):
                        geom = H.conv_geom(Rg, (Do, Ho, Wo), (kd, kh, kw), (1, 1, 1), (0, 0, 0), transposed=True)
                        gemm(P(gy), ldg, P(wd), Ci, 1, P(gxp), Ci, B * Rg[0] * Rg[1] * Rg[2], Ci, T * Co, Co, geom)
                    else:          # strided (MMVit2's down-sampling convs): a pad-0 strided conv of the padded grid, by parity classes
                        _dgrad_parity_classes(gy, ldg, wd, gxp, B, Rg, (Do, Ho, Wo), (kd, kh, kw), stride, (0, 0, 0), Ci, Co)
                    check(lib().corrif_pad_fold(P(gxp), P(gx), Ci, B, Di, Hi, Wi, Ci, stream()), "corrif_pad_fold")
                elif stride != (1, 1, 1):
                    _dgrad_parity_classes(gy, ldg, wd, gx, B, (Di, Hi, Wi), (Do, Ho, Wo), (kd, kh, kw), stride, pad, Ci, Co)
                else:
                    geom = H.conv_geom((Di, Hi, Wi), (Do, Ho, Wo), (kd, kh, kw), stride, pad, transposed=True)
                    st, bst = _bwd_stats_request(ctx.bwd_stats, ctx.grad_link, Min, Ci, dev, x, lda)
                    gemm(P(gy), ldg, P(wd), Ci, 1, P(gx), Ci, Min, Ci, T * Co, Co, geom, stats=st, bstats=bst)
        small11 = is_gemm and lib().corrif_conv1x1_small_supported(Ci, Co)
        if small11 and ctx.needs_input_grad[1]:      # weight and bias gradient in one streaming pass
            gw = torch.empty(weight.shape, dtype=torch.float32, device=dev)
            gb = torch.empty(Co, dtype=torch.float32, device=dev) if has_bias else None
            ws = _ws(lib().corrif_conv1x1_small_workspace(M, Ci, Co), dev)
            check(lib().corrif_conv1x1_small_wgrad(P(x), lda, P(gy), ldg, P(gw), P(gb), P(ws), M, Ci, Co, stream()), "corrif_conv1x1_small_wgrad")
            gx = _finish_link(ctx.grad_link, gx)
            return gx, gw, gb, None, None, None, None, None, None, None, None, None, None
        side = None
        if SIDE_WGRAD and ctx.side_ok and ctx.needs_input_grad[1] and weight.grad is None and not torch.cuda.is_current_stream_capturing():
            side = _side_begin((gy, x))
        with torch.cuda.stream(side) if side is not None else _NullCtx():
            gw, gb = ConvFn._weight_grads(ctx, x, weight, gy, M, ldg, dev)
        gx = _finish_link(ctx.grad_link, gx)
        return gx, gw, gb, None, None, None, None, None, None, None, None, None, None

    @staticmethod
    def _weight_grads(ctx, x, weight, gy, M, ldg, dev):
        stride, pad, replicate, has_bias, (B, Di, Hi, Wi), (Do, Ho, Wo), stem, is_gemm, lda, batch_pitch = ctx.cfg
        Co, Ci, kd, kh, kw = weight.shape
        T = kd * kh * kw
        gw = gb = None
        if ctx.needs_input_grad[1]:
            if stem and ctx.stem_k:
                gwp = torch.empty((Co, 148), dtype=torch.float32, device=dev)
                stem_wgrad(x, batch_pitch, gy, ldg, gwp, B, Di, Hi, Wi)
                gw = repack(gwp, weight.shape, Co, 1, T, 2, 148)
            elif stem:
                Kp = (T + 3) // 4 * 4
                gwp = torch.empty((Co, Kp), dtype=torch.float32, device=dev)
                geom = H.conv_geom((Do, Ho, Wo), (Di, Hi, Wi), (kd, kh, kw), stride, pad, clamp=replicate, ntaps=T, src_batch_pitch=batch_pitch)
                wgrad(P(gy), ldg, P(x), 1, 1, P(gwp), Kp, M, Co, Kp, geom, dev)
                gw = repack(gwp, weight.shape, Co, 1, T, 2, Kp)
            elif T == 1:
                gw = torch.empty(weight.shape, dtype=torch.float32, device=dev)
                geom = H.gemm_geom() if is_gemm else H.conv_geom((Do, Ho, Wo), (Di, Hi, Wi), (1, 1, 1), stride, pad)
                wgrad(P(gy), ldg, P(x), lda, Ci, P(gw), Ci, M, Co, Ci, geom, dev)
            else:
                gwp = torch.empty((Co, T * Ci), dtype=torch.float32, device=dev)
                if USE_PATCH and (kd, kh, kw) == (3, 3, 3) and stride == (1, 1, 1) and pad == (1, 1, 1) and lib().corrif_conv3_patch_wgrad_slots(Ci, Co):
                    conv3_patch_wgrad(P(x), lda, P(gy), ldg, P(gwp), B, (Di, Hi, Wi), (Do, Ho, Wo), Ci, Co, replicate, dev)
                else:
                    geom = H.conv_geom((Do, Ho, Wo), (Di, Hi, Wi), (kd, kh, kw), stride, pad, clamp=replicate)
                    wgrad(P(gy), ldg, P(x), lda, Ci, P(gwp), T * Ci, M, Co, T * Ci, geom, dev)
                gw = repack(gwp, weight.shape, Co, Ci, T, 2, T * Ci)
        if has_bias and ctx.needs_input_grad[2]:
            gb = col_sum(gy, M, ldg, Co)
        return gw, gb


TAP_STATS = {"epilogue": 0, "added": 0, "late": 0}      # grad_tap bookkeeping (tests): absorbed by a GEMM epilogue / own add pass / missed


def _finish_link(link, gx):
    """grad_tap bookkeeping at the end of ConvFn.backward: a tapped gradient that no kernel epilogue absorbed is added here; the
    link is marked done so that a tap firing later falls back to ordinary autograd accumulation."""
    if link is not None:
        while link.get("gs") and gx is not None:
            gx = add(gx, link["gs"].pop().contiguous())
            TAP_STATS["added"] += 1
        link["done"] = True
    return gx


class GradTapFn(Function):
    """Identity on a tensor that has another consumer whose backward can absorb this use's gradient (the Bottleneck input feeds the
    residual add / the downsample convolution AND conv1; a layer output also feeds the encoder's adapt convolution): backward parks
    the gradient in `link` for conv1's data-gradient GEMM epilogue and returns nothing, which removes the separate gradient-
    accumulation pass (read two tensors, write one) autograd would otherwise run per extra consumer.  The tap's node is created after
    conv1's, so the engine runs it first; if it ever ran after conv1's backward (`link["done"]`) it simply returns the gradient to
    autograd."""

    @staticmethod
    def forward(ctx, x, link):
        ctx.link = link
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.link.get("done"):
            TAP_STATS["late"] += 1
            ctx.link["late"] = True           # the absorbing GEMM's epilogue did not see this gradient (see _bwd_stats_request)
            return g, None
        ctx.link.setdefault("gs", []).append(g)
        return None, None


def grad_tap(x, link):
    return GradTapFn.apply(x, link)


def conv3d(x, weight, bias=None, stride=(1, 1, 1), pad=(0, 0, 0), replicate=False, out=None, act=ACT_NONE, stats=None, grad_link=None,
           side_wgrad=False, bwd_stats=None, bcast=None):
    """stats: None, or a dict {"G": groups, "relu": bool}: ask the GEMM epilogue for the statistics partials of the norm that follows;
    on success the dict gains "part" / "chunks" / "rpg" (pass it to batch_norm / relu_instnorm as `pre`)."""
    if bcast is not None and not (_patch_cc(tuple(weight.shape[2:]), tuple(stride), tuple(pad), weight.shape[1], weight.shape[0]) and act == ACT_NONE):
        bcast = None              # only the patch kernel's epilogue can absorb the depth-class add (bcast["done"] stays False: the caller adds)
    if bcast is None and stats is not None and stats.get("after_bcast"):
        stats = None              # the statistics must see the sum: without the fused add they are taken by the norm's own pass
    if bcast is None and act == ACT_NONE and grad_link is None and _fast():
        y, part, chunks, rpg = tl().conv3d_fwd(x, weight, bias, list(stride), list(pad), bool(replicate), out,
                                                stats["G"] if stats is not None else 0, bool(stats["relu"]) if stats is not None else False)
        if stats is not None and part.numel():
            stats["part"], stats["chunks"], stats["rpg"] = part, chunks, rpg
        return y
    return ConvFn.apply(x, weight, bias, tuple(stride), tuple(pad), bool(replicate), out, act, stats, grad_link, bool(side_wgrad), bwd_stats, bcast)


class SplitWeightFn(Function):
    """(w[:, :cs], w[:, cs:]) of a convolution weight [Co, C, kd, kh, kw] as two contiguous tensors (strided copy kernel); backward
    reassembles the two gradients into one tensor of the parameter's shape."""

    @staticmethod
    def forward(ctx, w, cs):
        Co, C = w.shape[:2]
        T = w.shape[2] * w.shape[3] * w.shape[4]
        w = w.contiguous()
        a = torch.empty((Co, cs) + tuple(w.shape[2:]), dtype=torch.float32, device=w.device)
        b = torch.empty((Co, C - cs) + tuple(w.shape[2:]), dtype=torch.float32, device=w.device)
        check(lib().corrif_copy2d(w.data_ptr(), C * T, P(a), cs * T, Co, cs * T, 0, stream()), "corrif_copy2d")
        check(lib().corrif_copy2d(w.data_ptr() + 4 * cs * T, C * T, P(b), (C - cs) * T, Co, (C - cs) * T, 0, stream()), "corrif_copy2d")
        ctx.cfg = (tuple(w.shape), cs, T)
        return a, b

    @staticmethod
    def backward(ctx, ga, gb):
        shape, cs, T = ctx.cfg
        Co, C = shape[:2]
        gw = torch.empty(shape, dtype=torch.float32, device=(ga if ga is not None else gb).device)
        if ga is None or gb is None:
            check(lib().corrif_fill(gw.data_ptr(), gw.numel(), 0.0, stream()), "corrif_fill")
        for g, off, n in ((ga, 0, cs), (gb, cs, C - cs)):
            if g is None:
                continue
            g = g.contiguous()
            check(lib().corrif_copy2d(P(g), n * T, gw.data_ptr() + 4 * off * T, C * T, Co, n * T, 0, stream()), "corrif_copy2d")
        return gw, None


def split_weight(w, cs):
    return SplitWeightFn.apply(w, cs)


class DepthBcastAddFn(Function):
    """y[b, d] += ys[b, cls(d)] in place (corrif_depth_bcast_add): `ys` lives on the compact depth grid of 3 * Ds slices (three depth
    classes per block of slices that share a nearest-neighbour source), see Decoder_fuse.forward.  Backward: the gradient passes
    through to y and is class-reduced for ys."""

    @staticmethod
    def forward(ctx, y, ys, fused=False):
        """fused: the convolution that produced y already added ys in its epilogue (CorrifConv3Patch.add_src): autograd wiring only"""
        B, D, Hh, W, C = y.shape
        yv, _, ldy = rows_view(y)
        sv, _, lds = rows_view(ys)
        Ds = ys.shape[1] // 3
        assert yv is y and tuple(ys.shape) == (B, 3 * Ds, Hh, W, C) and D >= 2 * Ds
        if not fused:
            check(lib().corrif_depth_bcast_add(P(y), ldy, P(sv), lds, B, D, Hh * W, C, Ds, stream()), "corrif_depth_bcast_add")
        ctx.mark_dirty(y)
        ctx.cfg = (Ds, tuple(ys.shape))
        return y

    @staticmethod
    def backward(ctx, g):
        Ds, sshape = ctx.cfg
        B, D, Hh, W, C = g.shape
        gv, _, ldg = rows_view(g)
        gs = torch.empty(sshape, dtype=torch.float32, device=g.device)
        check(lib().corrif_depth_class_reduce(P(gv), ldg, P(gs), C, B, D, Hh * W, C, Ds, stream()), "corrif_depth_class_reduce")
        return g, gs, None


def depth_bcast_add(y, ys, fused=False):
    return DepthBcastAddFn.apply(y, ys, bool(fused))


# --------------------------------------------------------------------------------------- grouped (stacked-modality) operators
# The reference runs the SAME Encoder three times, once per modality (mmvit4.py:442-447): every layer exists as three twins with identical
# geometry and their own weights.  One launch per twin leaves the chip unevenly filled (e3 / e4 / e5 have 196 * k tiles of 128 rows: 0.77 of a
# wave of workgroups), so the three twins run as ONE grouped launch: activations stacked along the batch axis ("stack":
# [G*B, D, H, W, C], group g = samples [g*B, (g+1)*B)) or living side by side in a concat buffer ("cat": [B, D, H, W, G*C], group g =
# channels [g*C, (g+1)*C) - the early-fusion buffers the adapt convolutions fill in place), weights stacked into one [G, ...] operand per step.
def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def stack_nograd(tensors):
    ts = [t.contiguous() for t in tensors]
    out = torch.empty((len(ts),) + tuple(ts[0].shape), dtype=torch.float32, device=ts[0].device)
    check(lib().corrif_stack_groups(_ptr_array(ts), len(ts), P(out), ts[0].numel(), stream()), "corrif_stack_groups")
    return out


class StackParamsFn(Function):
    """G same-shaped parameter tensors -> one contiguous [G, ...] operand (one gather launch); backward hands each parameter its slice
    of the stacked gradient (a view: autograd adopts it, no copy)."""

    @staticmethod
    def forward(ctx, *ts):
        return stack_nograd(ts)

    @staticmethod
    def backward(ctx, g):
        return tuple(g[i] for i in range(g.shape[0]))


def stack_params(tensors):
    return StackParamsFn.apply(*tensors)


class GroupedConvFn(Function):
    """G twin nn.Conv3d (same geometry, own weights w[g], bias b[g]) on channels-last activations as ONE launch per GEMM: forward, data
    gradient and weight gradient run with Z = G.  Covers the encoder's layer types: 1x1x1 (stride 1 or (1,2,2)), 1x3x3 (stride 1 or
    (1,2,2), pad (0,1,1)), zero padding, Ci > 1."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, G, zin, zout, out, stats_req, grad_link, bwd_stats, side_ok=False):
        ctx.side_ok = side_ok
        _, Co, Ci, kd, kh, kw = w.shape
        T = kd * kh * kw
        x, _, lda = rows_view(x)
        if zin == "stack":
            B, (Di, Hi, Wi) = x.shape[0] // G, x.shape[1:4]
            assert x.shape[0] == G * B and x.shape[-1] == Ci and lda == Ci
            zA = B * Di * Hi * Wi * lda
        else:
            B, (Di, Hi, Wi) = x.shape[0], x.shape[1:4]
            assert x.shape[-1] == G * Ci and T == 1 and stride == (1, 1, 1)
            zA = Ci
        Do, Ho, Wo = (_out_size(Di, kd, stride[0], pad[0]), _out_size(Hi, kh, stride[1], pad[1]), _out_size(Wi, kw, stride[2], pad[2]))
        M = B * Do * Ho * Wo                                   # rows per group
        if out is None:
            out = torch.empty((G * B, Do, Ho, Wo, Co) if zout == "stack" else (B, Do, Ho, Wo, G * Co), dtype=torch.float32, device=x.device)
        y, _, ldc = rows_view(out)
        assert y is out, "conv output must be row-addressable"
        zC = M * ldc if zout == "stack" else Co
        is_gemm = T == 1 and stride == (1, 1, 1)
        wp = w if T == 1 else repack(w, (G, Co, T * Ci), G * Co, Ci, T, 0, T * Ci)
        geom = H.gemm_geom() if is_gemm else H.conv_geom((Do, Ho, Wo), (Di, Hi, Wi), (kd, kh, kw), stride, pad)
        st, zs = None, {"bias": Co}
        if stats_req is not None and Co > 16:
            chunks = (M + 63) // 64                            # BatchNorm statistics partials per group, from the epilogue
            part = torch.empty(G * Co * chunks * 2, dtype=torch.float64, device=x.device)
            st = (part.data_ptr(), M, 1 if stats_req["relu"] else 0)
            zs["stats"] = Co * chunks * 2
            stats_req["part"], stats_req["chunks"], stats_req["rpg"] = part, chunks, M
        gemm(P(x), lda, P(wp), T * Ci, 0, P(y), ldc, M, Co, T * Ci, Ci, geom, bias=P(b) if b is not None else None, stats=st,
             Z=G, sA=(zA, 0), sB=(Co * T * Ci, 0), sC=(zC, 0), zs=zs)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, b is not None, G, zin, zout, B, (Di, Hi, Wi), (Do, Ho, Wo), is_gemm, lda, zA)
        ctx.grad_link, ctx.bwd_stats = grad_link, bwd_stats
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, pad, has_bias, G, zin, zout, B, (Di, Hi, Wi), (Do, Ho, Wo), is_gemm, lda, zA = ctx.cfg
        _, Co, Ci, kd, kh, kw = w.shape
        T = kd * kh * kw
        gy, _, ldg = rows_view(gy)
        dev = gy.device
        M, Min = B * Do * Ho * Wo, B * Di * Hi * Wi
        zG = M * ldg if zout == "stack" else Co
        assert ldg == (Co if zout == "stack" else G * Co)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if zin == "stack":
                gx, ldx, zX = torch.empty((G * B, Di, Hi, Wi, Ci), dtype=torch.float32, device=dev), Ci, Min * Ci
            else:
                gx, ldx, zX = torch.empty((B, Di, Hi, Wi, G * Ci), dtype=torch.float32, device=dev), G * Ci, Ci
            if is_gemm:
                adds, link = [], ctx.grad_link
                while link is not None and link.get("gs") and len(adds) < 2:      # gradients parked by the other consumers (grad_tap)
                    ga, _, ld_a = rows_view(link["gs"].pop())
                    adds.append((ga, ld_a))
                    TAP_STATS["epilogue"] += G               # one tapped gradient per twin module
                adds += [(None, 0)] * (2 - len(adds))
                st, bst, zs = _bwd_stats_request_g(ctx.bwd_stats, link, Min, Ci, G, dev, x, lda) if zin == "stack" else (None, None, {})
                if adds[0][0] is not None:
                    zs["add"] = Min * adds[0][1]
                if adds[1][0] is not None:
                    zs["add2"] = Min * adds[1][1]
                gemm(P(gy), ldg, P(w), Ci, 1, P(gx), ldx, Min, Ci, Co, Co, H.gemm_geom(), addend=P(adds[0][0]) if adds[0][0] is not None else None,
                     ld_add=adds[0][1], addend2=P(adds[1][0]) if adds[1][0] is not None else None, ld_add2=adds[1][1], stats=st, bstats=bst,
                     Z=G, sA=(zG, 0), sB=(Co * Ci, 0), sC=(zX, 0), zs=zs)
            else:
                if T == 1:
                    wd = w
                else:
                    wd = torch.empty((G, T * Co, Ci), dtype=torch.float32, device=dev)
                    for gi in range(G):                                            # tap-major data-gradient weights per group
                        check(lib().corrif_weight_repack(P(w[gi]), P(wd[gi]), Co, Ci, T, 1, T * Ci, stream()), "corrif_weight_repack")
                if stride != (1, 1, 1):
                    _dgrad_parity_classes(gy, ldg, wd, gx, B, (Di, Hi, Wi), (Do, Ho, Wo), (kd, kh, kw), stride, pad, Ci, Co, G=G, zG=zG, zX=zX)
                else:
                    geom = H.conv_geom((Di, Hi, Wi), (Do, Ho, Wo), (kd, kh, kw), stride, pad, transposed=True)
                    st, bst, zs = _bwd_stats_request_g(ctx.bwd_stats, ctx.grad_link, Min, Ci, G, dev, x, lda)
                    gemm(P(gy), ldg, P(wd), Ci, 1, P(gx), Ci, Min, Ci, T * Co, Co, geom, stats=st, bstats=bst,
                         Z=G, sA=(zG, 0), sB=(T * Co * Ci, 0), sC=(zX, 0), zs=zs)
        # weight / bias gradients are off the backward's critical path: on a side stream they keep the matrix pipe busy under the HBM-bound
        # BatchNorm backward passes that follow on the main stream (the grouped encoder has no sibling branch streams to fill them)
        side = None
        if SIDE_WGRAD_GROUPED and ctx.side_ok and ctx.needs_input_grad[1] and not torch.cuda.is_current_stream_capturing():
            side = _side_begin((gy, x, w))
        with torch.cuda.stream(side) if side is not None else _NullCtx():
            if ctx.needs_input_grad[1]:
                gw = torch.empty(w.shape, dtype=torch.float32, device=dev)
                if T == 1:
                    geom = H.gemm_geom() if is_gemm else H.conv_geom((Do, Ho, Wo), (Di, Hi, Wi), (1, 1, 1), stride, pad)
                    wgrad(P(gy), ldg, P(x), lda, Ci, P(gw), Ci, M, Co, Ci, geom, dev, Z=G, sA=(zG, 0), sB=(zA, 0), sC=(Co * Ci, 0))
                else:
                    gwp = torch.empty((G, Co, T * Ci), dtype=torch.float32, device=dev)
                    geom = H.conv_geom((Do, Ho, Wo), (Di, Hi, Wi), (kd, kh, kw), stride, pad)
                    wgrad(P(gy), ldg, P(x), lda, Ci, P(gwp), T * Ci, M, Co, T * Ci, geom, dev, Z=G, sA=(zG, 0), sB=(zA, 0), sC=(Co * T * Ci, 0))
                    check(lib().corrif_weight_repack(P(gwp), P(gw), G * Co, Ci, T, 2, T * Ci, stream()), "corrif_weight_repack")
            if has_bias and ctx.needs_input_grad[2]:
                gb = torch.empty((G, Co), dtype=torch.float32, device=dev)
                if zout == "stack":
                    ws = _ws(lib().corrif_norm_workspace_g(M, G, Co), dev)
                    check(lib().corrif_col_sum_g(P(gy), ldg, M, G, Co, P(gb), P(ws), stream()), "corrif_col_sum_g")
                else:                                     # concat layout: the column sums of the [M][G*Co] gradient ARE [G][Co]
                    ws = _ws(lib().corrif_col_sum_workspace(M, G * Co), dev)
                    check(lib().corrif_col_sum(P(gy), ldg, M, G * Co, P(gb), P(ws), stream()), "corrif_col_sum")
        gx = _finish_link(ctx.grad_link, gx)
        return gx, gw, gb, None, None, None, None, None, None, None, None, None, None


def _bwd_stats_request_g(bl, grad_link, rows, C, G, dev, xin, ldin):
    """grouped counterpart of _bwd_stats_request: returns (stats, bstats, zs strides) for ops.gemm"""
    if not BWD_STATS or bl is None or "x" not in bl or bl["rows"] != rows or bl["C"] != C or bl.get("G") != G or C <= 16:
        return None, None, {}
    if grad_link is not None and (grad_link.get("gs") or grad_link.get("late")):
        return None, None, {}
    chunks = (rows + 63) // 64
    part = torch.empty(G * C * chunks * 2, dtype=torch.float64, device=dev)
    bl["part"], bl["chunks"], bl["grad_link"] = part, chunks, grad_link
    zs = {"stats": C * chunks * 2, "bsx": rows * bl["ldx"], "bsy": rows * ldin, "bsstat": C}
    return ((part.data_ptr(), rows, 0), (P(bl["x"]), bl["ldx"], P(xin) if bl["relu_out"] else None, ldin, P(bl["mean"]), P(bl["rstd"])), zs)


def conv3d_grouped(x, weights, biases, stride, pad, zin="stack", zout="stack", out=None, stats=None, grad_link=None, bwd_stats=None):
    """weights / biases: the G twin modules' parameters (lists); see GroupedConvFn"""
    G = len(weights)
    if stats is None and grad_link is None and _fast():
        return tl().conv3d_grouped_fwd(x, list(weights), [] if biases[0] is None else list(biases), list(stride), list(pad),
                                       0 if zin == "stack" else 1, 0 if zout == "stack" else 1, out)
    w = stack_params(weights)
    b = stack_params(biases) if biases[0] is not None else None
    # the side-stream weight gradient is only safe when autograd will ADOPT the produced gradients (no accumulation kernel on another stream)
    side_ok = all(t.grad is None for t in weights) and all(t is None or t.grad is None for t in biases)
    return GroupedConvFn.apply(x, w, b, tuple(stride), tuple(pad), G, zin, zout, out, stats, grad_link, bwd_stats, side_ok)


class GroupedBatchNormFn(Function):
    """G twin nn.BatchNorm3d on activations stacked along the batch axis: group g = rows [g*rpg, (g+1)*rpg) with its own statistics,
    affine parameters gamma[g], beta[g] and running buffers.  Same arithmetic per group as BatchNormFn."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_means, running_vars, residual, flags, training, momentum, eps, out, pre, bwd_link, G):
        x, rows, ldx = rows_view(x)
        C = x.shape[-1]
        rpg = rows // G
        dev = x.device
        rstd = torch.empty((G, C), dtype=torch.float32, device=dev)
        if training:
            mean = torch.empty((G, C), dtype=torch.float32, device=dev)
            rm, rv = _ptr_array(running_means), _ptr_array(running_vars)
            if pre is not None and "part" in pre and pre["rpg"] == rpg and bool(pre["relu"]) == bool(flags & NORM_RELU_IN):
                check(lib().corrif_norm_stats_finalize_g(P(pre["part"]), pre["chunks"], G, C, rpg, eps, P(mean), P(rstd), rm, rv, momentum,
                                                         stream()), "corrif_norm_stats_finalize_g")
            else:
                ws = _ws(lib().corrif_norm_workspace_g(rpg, G, C), dev)
                check(lib().corrif_norm_stats_g(P(x), ldx, rpg, G, C, flags, eps, P(mean), P(rstd), rm, rv, momentum, P(ws), stream()),
                      "corrif_norm_stats_g")
        else:
            mean = stack_nograd(running_means)
            var = stack_nograd(running_vars)
            check(lib().corrif_norm_eval_rstd(P(var), eps, P(rstd), G * C, stream()), "corrif_norm_eval_rstd")
        ldr = 0
        if residual is not None:
            residual, _, ldr = rows_view(residual)
        if out is None:
            out = torch.empty(x.shape, dtype=torch.float32, device=dev)
        y, _, ldy = rows_view(out)
        assert y is out
        check(lib().corrif_norm_apply_g(P(x), ldx, P(mean), P(rstd), P(gamma), P(beta), P(residual), ldr, P(y), ldy, rpg, G, C, flags, C,
                                        stream()), "corrif_norm_apply_g")
        ctx.save_for_backward(x, mean, rstd, gamma, y if (flags & NORM_RELU_OUT) else None)
        ctx.cfg = (flags, not training, residual is not None, G)
        ctx.bwd_link = None
        if bwd_link is not None and training and not (flags & NORM_RELU_IN) and ldy == C:
            bwd_link.update(x=x, ldx=ldx, relu_out=bool(flags & NORM_RELU_OUT), mean=mean, rstd=rstd, rows=rpg, C=C, G=G)
            ctx.bwd_link = bwd_link
        return out

    @staticmethod
    def backward(ctx, gy):
        x, mean, rstd, gamma, y = ctx.saved_tensors
        flags, frozen, has_res, G = ctx.cfg
        x, rows, ldx = rows_view(x)
        gy, _, ldg = rows_view(gy)
        C = x.shape[-1]
        rpg = rows // G
        dev = x.device
        ldy = 0
        if y is not None:
            y, _, ldy = rows_view(y)
        gx = torch.empty(x.shape, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        gres = torch.empty(x.shape, dtype=torch.float32, device=dev) if (has_res and ctx.needs_input_grad[5]) else None
        ggamma = torch.empty((G, C), dtype=torch.float32, device=dev)
        gbeta = torch.empty((G, C), dtype=torch.float32, device=dev)
        bl = ctx.bwd_link
        if bl is not None and "part" in bl and not frozen and not (bl.get("grad_link") or {}).get("late") and ldg == C:
            ws = torch.empty(2 * G * C + 16, dtype=torch.float32, device=dev)
            check(lib().corrif_norm_bwd_pre_g(P(gy), ldg, P(y), ldy, P(x), ldx, P(mean), P(rstd), P(gamma), P(gx), C, P(gres), C, P(ggamma),
                                              P(gbeta), rpg, G, C, flags, C, P(bl.pop("part")), bl["chunks"], P(ws), stream()),
                  "corrif_norm_bwd_pre_g")
            NORM_BWD_STATS["epilogue"] += G
        else:
            ws = _ws(lib().corrif_norm_workspace_g(rpg, G, C), dev)
            check(lib().corrif_norm_bwd_g(P(gy), ldg, P(y), ldy, P(x), ldx, P(mean), P(rstd), P(gamma), P(gx), C, P(gres), C, P(ggamma),
                                          P(gbeta), rpg, G, C, flags, 1 if frozen else 0, C, P(ws), stream()), "corrif_norm_bwd_g")
            NORM_BWD_STATS["pass"] += G
        _drop_link(bl)
        return gx, ggamma, gbeta, None, None, gres, None, None, None, None, None, None, None, None


def batch_norm_grouped(x, gammas, betas, running_means, running_vars, residual=None, relu_in=False, relu_out=False, training=True,
                       momentum=0.1, eps=1e-5, out=None, pre=None, bwd_link=None):
    flags = (NORM_RELU_IN if relu_in else 0) | (NORM_RELU_OUT if relu_out else 0)
    if not training and out is None and _fast():
        return tl().batch_norm_grouped_eval(x, list(gammas), list(betas), list(running_means), list(running_vars), residual, flags, eps)
    return GroupedBatchNormFn.apply(x, stack_params(gammas), stack_params(betas), list(running_means), list(running_vars), residual, flags,
                                    training, momentum, eps, out, pre, bwd_link, len(gammas))


class CatBatchInPlaceFn(Function):
    """torch.cat(dim=0) whose parts were written in place into sample ranges of `buf` by their producers (the three stems)."""

    @staticmethod
    def forward(ctx, holder, *parts):
        ctx.ns = [p.shape[0] for p in parts]
        buf = holder[0]
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, g):
        outs, lo = [], 0
        for n in ctx.ns:
            outs.append(g[lo:lo + n])
            lo += n
        return (None,) + tuple(outs)


def cat_batch_inplace(buf, *parts):
    return CatBatchInPlaceFn.apply([buf], *parts)


class ResampleGroupsFn(Function):
    """The encoder's five `F.interpolate(x_l, size=(8,8,8), trilinear, align_corners=True)` (mmvit4.py:187-191) for G modalities whose x_l
    live side by side in a concat buffer [B, D, H, W, G*c]: group g is resampled into channels [off, off+c) of samples [g*B, (g+1)*B) of
    the stacked cube [G*B, 8, 8, 8, Ctot].  Backward writes the gradient of the whole concat buffer (one slice per group)."""

    @staticmethod
    def forward(ctx, x, cube, off, c, G):
        x, _, ldx = rows_view(x)
        B, Di, Hi, Wi, _ = x.shape
        Do, Ho, Wo, Ct = cube.shape[1:]
        assert cube.is_contiguous() and cube.shape[0] == G * B and x.shape[-1] == G * c
        for g in range(G):
            check(lib().corrif_trilinear_fwd(x.data_ptr() + 4 * g * c, ldx, cube.data_ptr() + 4 * (g * B * Do * Ho * Wo * Ct + off), Ct,
                                             B, c, Di, Hi, Wi, Do, Ho, Wo, stream()), "corrif_trilinear_fwd")
        ctx.cfg = (B, Di, Hi, Wi, off, c, G)
        return cube[..., off:off + c]

    @staticmethod
    def backward(ctx, g):
        B, Di, Hi, Wi, off, c, G = ctx.cfg
        g, _, ldg = rows_view(g)
        Do, Ho, Wo = g.shape[1:4]
        gx = torch.empty((B, Di, Hi, Wi, G * c), dtype=torch.float32, device=g.device)
        for gi in range(G):
            check(lib().corrif_trilinear_bwd(g.data_ptr() + 4 * gi * B * Do * Ho * Wo * ldg, ldg, gx.data_ptr() + 4 * gi * c, G * c,
                                             B, c, Di, Hi, Wi, Do, Ho, Wo, stream()), "corrif_trilinear_bwd")
        return gx, None, None, None, None


def resample_groups_to_cube(x, cube, off, c, G):
    return ResampleGroupsFn.apply(x, cube, off, c, G)


# --------------------------------------------------------------------------------------- linear
class LinearFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x, M, lda = rows_view(x)
        N, K = weight.shape
        y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
        gemm(P(x), lda, P(weight), K, 0, P(y), N, M, N, K, K, H.gemm_geom(), bias=P(bias) if bias is not None else None)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.lda = lda
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        N, K = weight.shape
        gy, M, ldg = rows_view(gy)
        dev = gy.device
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(x.shape, dtype=torch.float32, device=dev)
            gemm(P(gy), ldg, P(weight), K, 1, P(gx), K, M, K, N, N, H.gemm_geom())
        if ctx.needs_input_grad[1]:
            gw = torch.empty(weight.shape, dtype=torch.float32, device=dev)
            wgrad(P(gy), ldg, P(x), ctx.lda, K, P(gw), K, M, N, K, H.gemm_geom(), dev)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = col_sum(gy, M, ldg, N)
        return gx, gw, gb


def linear(x, weight, bias):
    if _fast():
        return tl().linear_fwd(x, weight, bias)
    return LinearFn.apply(x, weight, bias)


# --------------------------------------------------------------------------------------- BatchNorm / InstanceNorm
def _norm_ws(rows_per_group, G, C, dev):
    return _ws(lib().corrif_norm_workspace(rows_per_group, G, C), dev)


NORM_BWD_STATS = {"epilogue": 0, "pass": 0}      # BatchNorm backwards whose reduction came from a GEMM epilogue / ran as its own pass (tests)


def _drop_link(bl):
    """The backward-statistics link of a BatchNorm (a plain dict reachable from the autograd nodes' ctx objects) holds the layer's input
    activations.  Saved tensors are released by the backward pass, Python attributes of a ctx are not: a caller that keeps `loss` alive
    into the next step (`loss = criterion(...)` ... `loss.item()` as F4_TRAIN.py:58-64 does) would keep every BatchNorm input of the
    previous step alive with it (+35 GB at B = 32, 8 bands, 256^2).  Called at the end of the BatchNorm's backward, when nothing reads
    the link any more."""
    if bl is not None:
        for k in ("x", "mean", "rstd", "part"):
            bl.pop(k, None)


class BatchNormFn(Function):
    """y = act_out(gamma * (x' - mean) * rstd + beta + residual), x' = relu(x) if relu_in.  Statistics over all rows."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, flags, training, momentum, eps, out, pre, bwd_link=None):
        x, rows, ldx = rows_view(x)
        C = x.shape[-1]
        dev = x.device
        if training:
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            rstd = torch.empty(C, dtype=torch.float32, device=dev)
            if pre is not None and "part" in pre and pre["rpg"] == rows and pre["G"] == 1 and bool(pre["relu"]) == bool(flags & NORM_RELU_IN):
                check(lib().corrif_norm_stats_finalize(P(pre["part"]), pre["chunks"], 1, C, rows, eps, P(mean), P(rstd), P(running_mean),
                                                       P(running_var), momentum, stream()), "corrif_norm_stats_finalize")
            else:
                ws = _norm_ws(rows, 1, C, dev)
                check(lib().corrif_norm_stats(P(x), ldx, rows, 1, C, flags, eps, P(mean), P(rstd), P(running_mean), P(running_var),
                                              momentum, P(ws), stream()), "corrif_norm_stats")
        else:
            mean = running_mean
            rstd = torch.empty(C, dtype=torch.float32, device=dev)
            check(lib().corrif_norm_eval_rstd(P(running_var), eps, P(rstd), C, stream()), "corrif_norm_eval_rstd")
        ldr = 0
        if residual is not None:
            residual, _, ldr = rows_view(residual)
        if out is None:
            out = torch.empty(x.shape, dtype=torch.float32, device=dev)
        y, _, ldy = rows_view(out)
        assert y is out
        check(lib().corrif_norm_apply(P(x), ldx, P(mean), P(rstd), P(gamma), P(beta), P(residual), ldr, P(y), ldy, rows, 1, C, flags,
                                      stream()), "corrif_norm_apply")
        ctx.save_for_backward(x, mean, rstd, gamma, y if (flags & NORM_RELU_OUT) else None)
        ctx.cfg = (flags, not training, residual is not None)
        ctx.bwd_link = None
        if bwd_link is not None and training and not (flags & NORM_RELU_IN) and ldy == C:
            # whoever consumes `out` may compute this layer's backward statistics in its data-gradient epilogue (_bwd_stats_request)
            bwd_link.update(x=x, ldx=ldx, relu_out=bool(flags & NORM_RELU_OUT), mean=mean, rstd=rstd, rows=rows, C=C)
            ctx.bwd_link = bwd_link
        return out

    @staticmethod
    def backward(ctx, gy):
        x, mean, rstd, gamma, y = ctx.saved_tensors
        flags, frozen, has_res = ctx.cfg
        x, rows, ldx = rows_view(x)
        gy, _, ldg = rows_view(gy)
        C = x.shape[-1]
        dev = x.device
        ldy = 0
        if y is not None:
            y, _, ldy = rows_view(y)
        gx = torch.empty(x.shape, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        gres = torch.empty(x.shape, dtype=torch.float32, device=dev) if (has_res and ctx.needs_input_grad[5]) else None
        ggamma = torch.empty(C, dtype=torch.float32, device=dev)
        gbeta = torch.empty(C, dtype=torch.float32, device=dev)
        bl = ctx.bwd_link
        if bl is not None and "part" in bl and not frozen and not (bl.get("grad_link") or {}).get("late") and ldg == C:
            # the reduction (sum g', sum g' xhat) came out of the epilogue of the data-gradient GEMM that produced gy
            ws = torch.empty(2 * C + 16, dtype=torch.float32, device=dev)
            check(lib().corrif_norm_bwd_pre(P(gy), ldg, P(y), ldy, P(x), ldx, P(mean), P(rstd), P(gamma), P(gx), C, P(gres), C, P(ggamma), P(gbeta),
                                            rows, C, flags, P(bl.pop("part")), bl["chunks"], P(ws), stream()), "corrif_norm_bwd_pre")
            NORM_BWD_STATS["epilogue"] += 1
        else:
            ws = _norm_ws(rows, 1, C, dev)
            check(lib().corrif_norm_bwd(P(gy), ldg, P(y), ldy, P(x), ldx, P(mean), P(rstd), P(gamma), P(gx), C, P(gres), C,
                                        P(ggamma), P(gbeta), rows, 1, C, flags, 1 if frozen else 0, P(ws), stream()), "corrif_norm_bwd")
            NORM_BWD_STATS["pass"] += 1
        _drop_link(bl)
        return gx, ggamma, gbeta, None, None, gres, None, None, None, None, None, None, None


def batch_norm(x, gamma, beta, running_mean, running_var, residual=None, relu_in=False, relu_out=False, training=True,
               momentum=0.1, eps=1e-5, out=None, pre=None, bwd_link=None):
    flags = (NORM_RELU_IN if relu_in else 0) | (NORM_RELU_OUT if relu_out else 0)
    if not training and _fast():
        return tl().batch_norm_eval(x, gamma, beta, running_mean, running_var, residual, flags, eps, out)
    return BatchNormFn.apply(x, gamma, beta, running_mean, running_var, residual, flags, training, momentum, eps, out, pre, bwd_link)


class ReluInstNormFn(Function):
    """InstanceNorm3d(relu(x)), affine=False, eps 1e-5: statistics per (sample, channel)."""

    @staticmethod
    def forward(ctx, x, eps, out, pre):
        x, rows, ldx = rows_view(x)
        B, C = x.shape[0], x.shape[-1]
        rpg = rows // B
        dev = x.device
        mean = torch.empty(B * C, dtype=torch.float32, device=dev)
        rstd = torch.empty(B * C, dtype=torch.float32, device=dev)
        if pre is not None and "part" in pre and pre["rpg"] == rpg and pre["G"] == B and pre["relu"]:
            check(lib().corrif_norm_stats_finalize(P(pre["part"]), pre["chunks"], B, C, rpg, eps, P(mean), P(rstd), 0, 0, 0.0, stream()),
                  "corrif_norm_stats_finalize")
        else:
            ws = _norm_ws(rpg, B, C, dev)
            check(lib().corrif_norm_stats(P(x), ldx, rpg, B, C, NORM_RELU_IN, eps, P(mean), P(rstd), None, None, 0.0, P(ws), stream()),
                  "corrif_norm_stats")
        if out is None:
            out = torch.empty(x.shape, dtype=torch.float32, device=dev)
        y, _, ldy = rows_view(out)
        assert y is out
        check(lib().corrif_norm_apply(P(x), ldx, P(mean), P(rstd), None, None, None, 0, P(y), ldy, rpg, B, C, NORM_RELU_IN, stream()),
              "corrif_norm_apply")
        ctx.save_for_backward(x, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, mean, rstd = ctx.saved_tensors
        x, rows, ldx = rows_view(x)
        gy, _, ldg = rows_view(gy)
        B, C = x.shape[0], x.shape[-1]
        rpg = rows // B
        dev = x.device
        gx = torch.empty(x.shape, dtype=torch.float32, device=dev)
        ws = _norm_ws(rpg, B, C, dev)
        check(lib().corrif_norm_bwd(P(gy), ldg, None, 0, P(x), ldx, P(mean), P(rstd), None, P(gx), C, None, 0, None, None,
                                    rpg, B, C, NORM_RELU_IN, 0, P(ws), stream()), "corrif_norm_bwd")
        return gx, None, None, None


_NO_PART = {}


def relu_instnorm(x, eps=1e-5, out=None, pre=None):
    if _fast():
        if pre is not None and "part" in pre and pre["G"] == x.shape[0] and pre["relu"]:
            return tl().relu_instnorm_fwd(x, eps, out, pre["part"], pre["chunks"], pre["rpg"])
        key = x.device
        if key not in _NO_PART:
            _NO_PART[key] = torch.empty(0, dtype=torch.float64, device=x.device)
        return tl().relu_instnorm_fwd(x, eps, out, _NO_PART[key], 0, 0)
    return ReluInstNormFn.apply(x, eps, out, pre)


# --------------------------------------------------------------------------------------- LayerNorm
class LayerNormFn(Function):
    """pos given: returns (xs, y) = (x + pos, LN(x + pos))   (Transformer.forward `x = x + pos`, mmvit4.py:385)
    pos None : returns y = LN(x)."""

    @staticmethod
    def forward(ctx, x, pos, gamma, beta, eps):
        x = x.contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        dev = x.device
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=dev)
        rstd = torch.empty(rows, dtype=torch.float32, device=dev)
        xs, pos_rows = x, 0
        if pos is not None:
            pos = pos.contiguous()
            xs = torch.empty_like(x)
            pos_rows = pos.numel() // C
        check(lib().corrif_layernorm_fwd(P(x), P(pos), pos_rows, P(xs) if pos is not None else 0, P(gamma), P(beta), P(y), P(mean),
                                         P(rstd), rows, C, eps, stream()), "corrif_layernorm_fwd")
        ctx.save_for_backward(xs, mean, rstd, gamma)
        ctx.pos_shape = None if pos is None else (pos.shape, pos_rows)
        return (xs, y) if pos is not None else y

    @staticmethod
    def backward(ctx, *grads):
        xs, mean, rstd, gamma = ctx.saved_tensors
        gxs, gy = grads if ctx.pos_shape is not None else (None, grads[0])
        C = xs.shape[-1]
        rows = xs.numel() // C
        dev = xs.device
        gy = gy.contiguous()
        gx = torch.empty_like(xs)
        ggamma = torch.empty(C, dtype=torch.float32, device=dev)
        gbeta = torch.empty(C, dtype=torch.float32, device=dev)
        ws = _ws(lib().corrif_layernorm_workspace(rows, C), dev)
        check(lib().corrif_layernorm_bwd(P(gy), P(xs), P(mean), P(rstd), P(gamma), P(gx), P(ws), P(ggamma), P(gbeta), rows, C, stream()),
              "corrif_layernorm_bwd")
        gpos = None
        if ctx.pos_shape is not None:
            if gxs is not None:
                gxs = gxs.contiguous()
                check(lib().corrif_add(P(gx), P(gxs), P(gx), gx.numel(), stream()), "corrif_add")
            shape, pos_rows = ctx.pos_shape
            gpos = torch.empty(shape, dtype=torch.float32, device=dev)
            check(lib().corrif_sum_groups(P(gx), P(gpos), pos_rows * C, rows // pos_rows, stream()), "corrif_sum_groups")
        return gx, gpos, ggamma, gbeta, None


def layer_norm(x, gamma, beta, pos=None, eps=1e-5):
    """pos given: (x + pos, LN(x + pos)); else LN(x)"""
    return LayerNormFn.apply(x, pos, gamma, beta, eps)


# --------------------------------------------------------------------------------------- element-wise
class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        check(lib().corrif_add(P(a), P(b), P(y), a.numel(), stream()), "corrif_add")
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g


add = AddFn.apply


class GeluFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib().corrif_gelu_fwd(P(x), P(y), x.numel(), stream()), "corrif_gelu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(x)
        check(lib().corrif_gelu_bwd(P(g), P(x), P(gx), x.numel(), stream()), "corrif_gelu_bwd")
        return gx


gelu = GeluFn.apply


class _Philox:
    """Counter-based dropout stream: (seed, running offset).

    The seed follows torch's default generator: whenever `torch.initial_seed()` differs from the value seen at the previous
    reservation (`torch.manual_seed(...)` was called, as F2_MAIN-style scripts do for reproducibility), the stream is re-keyed
    from it and the offset restarts at 0; `rank_offset` (set by the data-parallel layer) decorrelates the ranks of one job.
    `ops.manual_seed(s)` pins an explicit seed instead (bench.py, tests)."""
    seed = None             # None: follow torch.initial_seed()
    offset = 0
    rank_offset = 0
    _torch_seed = None

    @classmethod
    def reserve(cls, n):
        if cls.seed is None or cls._torch_seed is not None:
            ts = torch.initial_seed()
            if ts != cls._torch_seed:
                cls._torch_seed = ts
                cls.seed = (ts * 0x9E3779B97F4A7C15 + 0x5EED + cls.rank_offset * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
                cls.offset = 0
        off = cls.offset
        cls.offset += (n + 3) // 4 * 4
        return cls.seed, off


def manual_seed(seed):
    """explicit dropout seed (detaches the stream from torch's generator until `follow_torch_seed()`)"""
    _Philox.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    _Philox._torch_seed = None
    _Philox.offset = 0


def follow_torch_seed(rank=0):
    """derive the dropout stream from torch.initial_seed() (+ the data-parallel rank) again"""
    _Philox.seed, _Philox._torch_seed, _Philox.rank_offset, _Philox.offset = None, None, int(rank), 0


class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p):
        x = x.contiguous()
        seed, off = _Philox.reserve(x.numel())
        y = torch.empty_like(x)
        check(lib().corrif_dropout(P(x), P(y), x.numel(), p, seed, off, stream()), "corrif_dropout")
        ctx.cfg = (p, seed, off)
        return y

    @staticmethod
    def backward(ctx, g):
        p, seed, off = ctx.cfg
        g = g.contiguous()
        gx = torch.empty_like(g)
        check(lib().corrif_dropout(P(g), P(gx), g.numel(), p, seed, off, stream()), "corrif_dropout")
        return gx, None


def dropout(x, p, training):
    if not training or p == 0.0:
        return x
    return DropoutFn.apply(x, p)


# --------------------------------------------------------------------------------------- attention core
class AttentionFn(Function):
    """softmax(q k^T * scale) (dropout) v for qkv [B, N, 3*heads*hd] laid out as reshape(B,N,3,heads,hd) (mmvit4.py:307-312).
    Scores are materialised ([B, heads, N, N]); q.k^T, p.v and the four backward products run on the MFMA GEMM kernels."""

    @staticmethod
    def forward(ctx, qkv, heads, p_drop, training):
        qkv = qkv.contiguous()
        B, N, C3 = qkv.shape
        C = C3 // 3
        hd = C // heads
        dev = qkv.device
        scale = hd ** -0.5
        Z = B * heads
        S = torch.empty((B, heads, N, N), dtype=torch.float32, device=dev)
        base = qkv.data_ptr()
        q_ptr, k_ptr, v_ptr = base, base + 4 * C, base + 8 * C
        gg = H.gemm_geom()
        # S[b,h] = Q K^T : A = q rows (pitch 3C), B = k rows [N][hd] (pitch 3C)
        gemm(q_ptr, C3, k_ptr, C3, 0, P(S), N, N, N, hd, hd, gg, Z=Z, Zi=heads, sA=(N * C3, hd), sB=(N * C3, hd), sC=(heads * N * N, N * N))
        drop = training and p_drop > 0.0
        Pd = S
        seed = off = 0
        if drop:                # softmax and attention dropout in one pass over the scores
            seed, off = _Philox.reserve(S.numel())
            Pd = torch.empty_like(S)
            check(lib().corrif_softmax_dropout_rows(P(S), P(Pd), Z * N, N, scale, p_drop, seed, off, stream()), "corrif_softmax_dropout_rows")
        else:
            check(lib().corrif_softmax_rows(P(S), Z * N, N, scale, stream()), "corrif_softmax_rows")
        out = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        # O[b,:,h,:] = P V : B operand is V as [K = tokens][N = hd] (layout 1, pitch 3C)
        gemm(P(Pd), N, v_ptr, C3, 1, P(out), C, N, hd, N, N, gg, Z=Z, Zi=heads, sA=(heads * N * N, N * N), sB=(N * C3, hd), sC=(N * C, hd))
        ctx.save_for_backward(qkv, S)
        ctx.cfg = (heads, p_drop, drop, seed, off, scale)
        return out

    @staticmethod
    def backward(ctx, go):
        qkv, S = ctx.saved_tensors
        heads, p_drop, drop, seed, off, scale = ctx.cfg
        B, N, C3 = qkv.shape
        C = C3 // 3
        hd = C // heads
        dev = qkv.device
        Z = B * heads
        go = go.contiguous()
        base = qkv.data_ptr()
        q_ptr, k_ptr, v_ptr = base, base + 4 * C, base + 8 * C
        dqkv = torch.empty_like(qkv)
        dbase = dqkv.data_ptr()
        dq_ptr, dk_ptr, dv_ptr = dbase, dbase + 4 * C, dbase + 8 * C
        gg = H.gemm_geom()
        Pd = S
        if drop:
            Pd = torch.empty_like(S)
            check(lib().corrif_dropout(P(S), P(Pd), S.numel(), p_drop, seed, off, stream()), "corrif_dropout")
        # dV[m, d] = sum_n P'[n, m] dO[n, d]
        wgrad(P(Pd), N, P(go), C, hd, dv_ptr, C3, N, N, hd, gg, dev, Z=Z, Zi=heads, sA=(heads * N * N, N * N), sB=(N * C, hd), sC=(N * C3, hd))
        # dP' = dO V^T  (reuse the P' buffer when it is a private copy)
        dP = Pd if drop else torch.empty_like(S)
        gemm(P(go), C, v_ptr, C3, 0, P(dP), N, N, N, hd, hd, gg, Z=Z, Zi=heads, sA=(N * C, hd), sB=(N * C3, hd), sC=(heads * N * N, N * N))
        if drop:
            check(lib().corrif_softmax_dropout_rows_bwd(P(S), P(dP), Z * N, N, scale, p_drop, seed, off, stream()), "corrif_softmax_dropout_rows_bwd")
        else:
            check(lib().corrif_softmax_rows_bwd(P(S), P(dP), Z * N, N, scale, stream()), "corrif_softmax_rows_bwd")
        # dQ = dS K ; dK = dS^T Q
        gemm(P(dP), N, k_ptr, C3, 1, dq_ptr, C3, N, hd, N, N, gg, Z=Z, Zi=heads, sA=(heads * N * N, N * N), sB=(N * C3, hd), sC=(N * C3, hd))
        wgrad(P(dP), N, q_ptr, C3, hd, dk_ptr, C3, N, N, hd, gg, dev, Z=Z, Zi=heads, sA=(heads * N * N, N * N), sB=(N * C3, hd), sC=(N * C3, hd))
        return dqkv, None, None, None


def flash_fwd(qkv, out, lse, mask, B, N, heads, scale, p_drop, seed, off):
    check(lib().corrif_flash_attn_fwd(P(qkv), P(out), P(lse), P(mask), B, N, heads, scale, p_drop, seed, off, stream()), "corrif_flash_attn_fwd")


def flash_bwd(qkv, out, lse, mask, go, dvec, dqkv, B, N, heads, scale, p_drop):
    check(lib().corrif_flash_attn_bwd(P(qkv), P(out), P(lse), P(mask), P(go), P(dvec), P(dqkv), B, N, heads, scale, p_drop, stream()),
          "corrif_flash_attn_bwd")


class FlashAttentionFn(Function):
    """The same attention core without the [B, heads, N, N] tensors (corrif_flash_attn_*): forward keeps one 64-key score tile in
    MFMA accumulators, backward recomputes the probabilities from qkv and the saved row log-sum-exp.  Same Philox indexing as
    AttentionFn, so both draw the identical dropout mask from one (seed, offset)."""

    @staticmethod
    def forward(ctx, qkv, heads, p_drop, training):
        qkv = qkv.contiguous()
        B, N, C3 = qkv.shape
        C = C3 // 3
        scale = (C // heads) ** -0.5
        drop = training and p_drop > 0.0
        seed = off = 0
        if drop:
            seed, off = _Philox.reserve(B * heads * N * N)
        out = torch.empty((B, N, C), dtype=torch.float32, device=qkv.device)
        lse = torch.empty((B * heads, N), dtype=torch.float32, device=qkv.device)
        mask = None
        if drop:         # keep bits, 1/32 of a score tensor: drawn once by the forward, read by the backward kernels
            mask = torch.empty(lib().corrif_flash_attn_mask_bytes(B, N, heads, p_drop) // 4, dtype=torch.int32, device=qkv.device)
        flash_fwd(qkv, out, lse, mask, B, N, heads, scale, p_drop if drop else 0.0, seed, off)
        ctx.save_for_backward(qkv, out, lse, mask)
        ctx.cfg = (heads, p_drop if drop else 0.0, scale)
        return out

    @staticmethod
    def backward(ctx, go):
        qkv, out, lse, mask = ctx.saved_tensors
        heads, p_drop, scale = ctx.cfg
        B, N, _ = qkv.shape
        go = go.contiguous()
        dqkv = torch.empty_like(qkv)
        dvec = torch.empty_like(lse)
        flash_bwd(qkv, out, lse, mask, go, dvec, dqkv, B, N, heads, scale, p_drop)
        return dqkv, None, None, None


FLASH_ATTENTION = True      # diagnostics / A-B: False materialises the score tensor (AttentionFn)


def attention(qkv, heads, p_drop, training):
    if FLASH_ATTENTION and lib().corrif_flash_attn_supported(qkv.shape[1], qkv.shape[2] // (3 * heads)):
        return FlashAttentionFn.apply(qkv, heads, p_drop, training)
    return AttentionFn.apply(qkv, heads, p_drop, training)


# --------------------------------------------------------------------------------------- inter-modal correlation
class InterCorrFn(Function):
    @staticmethod
    def forward(ctx, a, b, c):
        a, b, c = a.contiguous(), b.contiguous(), c.contiguous()
        B, S, C3 = a.shape
        C = C3 // 3
        outs = [torch.empty((B, S, C), dtype=torch.float32, device=a.device) for _ in range(3)]
        check(lib().corrif_intercorr_fwd(P(a), P(b), P(c), C3, P(outs[0]), P(outs[1]), P(outs[2]), C, B, S, C, stream()), "corrif_intercorr_fwd")
        ctx.save_for_backward(a, b, c)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g0, g1, g2):
        a, b, c = ctx.saved_tensors
        B, S, C3 = a.shape
        C = C3 // 3
        gs = [g.contiguous() if g is not None else torch.zeros((B, S, C), dtype=torch.float32, device=a.device) for g in (g0, g1, g2)]
        ds = [torch.empty_like(a) for _ in range(3)]
        check(lib().corrif_intercorr_bwd(P(a), P(b), P(c), C3, P(gs[0]), P(gs[1]), P(gs[2]), C, P(ds[0]), P(ds[1]), P(ds[2]), B, S, C, stream()),
              "corrif_intercorr_bwd")
        return tuple(ds)


inter_corr = InterCorrFn.apply


# --------------------------------------------------------------------------------------- pooling / resampling
class MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, D, Hh, W, C = x.shape
        Ho, Wo = (Hh - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, D, Ho, Wo, C), dtype=torch.float32, device=x.device)
        idx = torch.empty((B, D, Ho, Wo, C), dtype=torch.int8, device=x.device)
        check(lib().corrif_maxpool133_fwd(P(x), P(y), P(idx), B, D, Hh, W, C, stream()), "corrif_maxpool133_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (B, D, Hh, W, C)
        return y

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, D, Hh, W, C = ctx.shape
        g = g.contiguous()
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        check(lib().corrif_maxpool133_bwd(P(g), P(idx), P(gx), B, D, Hh, W, C, stream()), "corrif_maxpool133_bwd")
        return gx


maxpool133 = MaxPoolFn.apply


class ResampleFn(Function):
    @staticmethod
    def forward(ctx, x, size, mode, out):
        x, _, ldx = rows_view(x)
        B, Di, Hi, Wi, C = x.shape
        Do, Ho, Wo = size
        if out is None:
            out = torch.empty((B, Do, Ho, Wo, C), dtype=torch.float32, device=x.device)
        y, _, ldy = rows_view(out)
        assert y is out
        fn = lib().corrif_trilinear_fwd if mode == "trilinear" else lib().corrif_nearest_fwd
        check(fn(P(x), ldx, P(y), ldy, B, C, Di, Hi, Wi, Do, Ho, Wo, stream()), "corrif_%s_fwd" % mode)
        ctx.cfg = (mode, (B, Di, Hi, Wi, C), size)
        return out

    @staticmethod
    def backward(ctx, g):
        mode, (B, Di, Hi, Wi, C), (Do, Ho, Wo) = ctx.cfg
        g, _, ldg = rows_view(g)
        gx = torch.empty((B, Di, Hi, Wi, C), dtype=torch.float32, device=g.device)
        if mode == "trilinear" and TRILINEAR_SEPARABLE and ldg == C and Do * Ho * Wo >= 4 * Di * Hi * Wi and g.numel() >= (1 << 20):
            need = lib().corrif_trilinear_bwd_sep_workspace(B, C, Di, Hi, Wi, Do, Ho, Wo)
            if need >= 0:         # the decoder's x2 up-samplings: one axis per pass, every incoming gradient read once
                ws = torch.empty(need // 4, dtype=torch.float32, device=g.device)
                check(lib().corrif_trilinear_bwd_sep(P(g), P(gx), P(ws), B, C, Di, Hi, Wi, Do, Ho, Wo, stream()), "corrif_trilinear_bwd_sep")
                return gx, None, None, None
        fn = lib().corrif_trilinear_bwd if mode == "trilinear" else lib().corrif_nearest_bwd
        check(fn(P(g), ldg, P(gx), C, B, C, Di, Hi, Wi, Do, Ho, Wo, stream()), "corrif_%s_bwd" % mode)
        return gx, None, None, None


def trilinear(x, size, out=None):
    return ResampleFn.apply(x, tuple(size), "trilinear", out)


def nearest(x, size, out=None):
    return ResampleFn.apply(x, tuple(size), "nearest", out)


# --------------------------------------------------------------------------------------- concatenation
class CatChannelsFn(Function):
    """torch.cat(dim=channels) whose parts were written in place into channel slices of `buf` by their producers."""

    @staticmethod
    def forward(ctx, holder, *parts):
        ctx.widths = [p.shape[-1] for p in parts]
        buf = holder[0]
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, g):
        outs, o = [], 0
        for w in ctx.widths:
            outs.append(g[..., o:o + w])
            o += w
        return (None,) + tuple(outs)


def cat_channels(buf, *parts):
    return CatChannelsFn.apply([buf], *parts)


class CatChannelsCopyFn(Function):
    """torch.cat(parts, dim=channels) of channels-last tensors produced elsewhere (one strided copy per part);
    backward hands out channel-slice views of the gradient."""

    @staticmethod
    def forward(ctx, *parts):
        ctx.widths = [p.shape[-1] for p in parts]
        lead = parts[0].shape[:-1]
        out = torch.empty(lead + (sum(ctx.widths),), dtype=torch.float32, device=parts[0].device)
        rows, C, o = out.numel() // out.shape[-1], out.shape[-1], 0
        for p_ in parts:
            p_, _, ld = rows_view(p_)
            w = p_.shape[-1]
            check(lib().corrif_copy2d(P(p_), ld, out.data_ptr() + 4 * o, C, rows, w, 0, stream()), "corrif_copy2d")
            o += w
        return out

    @staticmethod
    def backward(ctx, g):
        outs, o = [], 0
        for w in ctx.widths:
            outs.append(g[..., o:o + w])
            o += w
        return tuple(outs)


def cat_channels_copy(*parts):
    return CatChannelsCopyFn.apply(*parts)


class SplitBatchFn(Function):
    """t[b0:b1], t[b1:b2], ... along the batch dimension (views); backward reassembles the gradient pieces with the copy kernel"""

    @staticmethod
    def forward(ctx, t, bounds):
        ctx.bounds, ctx.shape = list(bounds), t.shape
        return tuple(t[bounds[k]:bounds[k + 1]] for k in range(len(bounds) - 1))

    @staticmethod
    def backward(ctx, *gs):
        dev = next(g for g in gs if g is not None).device
        out = torch.empty(ctx.shape, dtype=torch.float32, device=dev)
        per = out[0].numel()
        for k, g in enumerate(gs):
            lo, n = ctx.bounds[k], ctx.bounds[k + 1] - ctx.bounds[k]
            dst = out[lo:lo + n]
            if g is None:
                dst.zero_()
            else:
                g = g.contiguous()
                check(lib().corrif_copy2d(P(g), per, P(dst), per, n, per, 0, stream()), "corrif_copy2d")
        return out, None


def split_batch(t, bounds):
    return SplitBatchFn.apply(t, bounds)


class CatBatchFn(Function):
    """torch.cat(parts, 0) for equally shaped sample blocks; backward hands out views of the gradient"""

    @staticmethod
    def forward(ctx, *parts):
        ns = [p.shape[0] for p in parts]
        out = torch.empty((sum(ns),) + tuple(parts[0].shape[1:]), dtype=torch.float32, device=parts[0].device)
        per = out[0].numel()
        lo = 0
        for p_, n in zip(parts, ns):
            p_ = p_.contiguous()
            check(lib().corrif_copy2d(P(p_), per, P(out[lo:lo + n]), per, n, per, 0, stream()), "corrif_copy2d")
            lo += n
        ctx.ns = ns
        return out

    @staticmethod
    def backward(ctx, g):
        outs, lo = [], 0
        for n in ctx.ns:
            outs.append(g[lo:lo + n])
            lo += n
        return tuple(outs)


cat_batch = CatBatchFn.apply


class CatTokensFn(Function):
    """torch.cat(dim=1) of [B, n_i, C] token blocks (mmvit4.py:515-521)."""

    @staticmethod
    def forward(ctx, *parts):
        parts = [p.contiguous() for p in parts]
        B, C = parts[0].shape[0], parts[0].shape[-1]
        ns = [p.shape[1] for p in parts]
        Nt = sum(ns)
        buf = torch.empty((B, Nt, C), dtype=torch.float32, device=parts[0].device)
        o = 0
        for p_, n in zip(parts, ns):
            check(lib().corrif_copy2d(P(p_), n * C, buf.data_ptr() + 4 * o * C, Nt * C, B, n * C, 0, stream()), "corrif_copy2d")
            o += n
        ctx.ns = ns
        return buf

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, Nt, C = g.shape
        outs, o = [], 0
        for n in ctx.ns:
            t = torch.empty((B, n, C), dtype=torch.float32, device=g.device)
            check(lib().corrif_copy2d(g.data_ptr() + 4 * o * C, Nt * C, P(t), n * C, B, n * C, 0, stream()), "corrif_copy2d")
            outs.append(t)
            o += n
        return tuple(outs)


cat_tokens = CatTokensFn.apply


# --------------------------------------------------------------------------------------- head / loss / metric
class HeadFn(Function):
    """final_conv (8 -> 3, 1x1x1, bias) + sigmoid; channels-last [B,1,224,224,8] -> NCDHW [B,3,1,224,224]."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        B = x.shape[0]
        HW = x.numel() // (B * 8)
        pred = torch.empty((B, 3, 1, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
        check(lib().corrif_head_fwd(P(x), P(weight), P(bias), P(pred), B, HW, stream()), "corrif_head_fwd")
        ctx.save_for_backward(x, weight, pred)
        return pred

    @staticmethod
    def backward(ctx, g):
        x, weight, pred = ctx.saved_tensors
        B = x.shape[0]
        HW = x.numel() // (B * 8)
        g = g.contiguous()
        gx = torch.empty_like(x)
        gw = torch.empty_like(weight)
        gb = torch.empty(3, dtype=torch.float32, device=x.device)
        ws = _ws(lib().corrif_head_workspace(B, HW), x.device)
        check(lib().corrif_head_bwd(P(g), P(pred), P(x), P(weight), P(gx), P(gw), P(gb), P(ws), B, HW, stream()), "corrif_head_bwd")
        return gx, gw, gb


head = HeadFn.apply


class BCEWithLogitsMeanFn(Function):
    """nn.BCEWithLogitsLoss()(pred, target) as called at F4_TRAIN.py:58-60."""

    @staticmethod
    def forward(ctx, pred, target):
        pred, target = pred.contiguous(), target.contiguous()
        n = pred.numel()
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred)
        ws = _ws(lib().corrif_bce_workspace(n), pred.device)
        check(lib().corrif_bce_logits_mean(P(pred), P(target), n, P(loss), P(dpred), P(ws), stream()), "corrif_bce_logits_mean")
        ctx.save_for_backward(dpred)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        out = torch.empty_like(dpred)
        check(lib().corrif_scale_dev(P(dpred), P(g.contiguous()), P(out), dpred.numel(), stream()), "corrif_scale_dev")
        return out, None


bce_with_logits_mean = BCEWithLogitsMeanFn.apply


def jaccard_all(y, y_pred, eps=1e-8):
    """returns a [3] tensor: Jaccard2, Jaccard, JaccardAndF1 (F5_JACCARD2.py:4-37) for y, y_pred of n elements"""
    y, y_pred = y.contiguous(), y_pred.contiguous()
    n = y.numel()
    out = torch.empty(3, dtype=torch.float32, device=y.device)
    ws = _ws(lib().corrif_jaccard_workspace(n), y.device)
    check(lib().corrif_jaccard(P(y), P(y_pred), n, eps, P(out), P(ws), stream()), "corrif_jaccard")
    return out
