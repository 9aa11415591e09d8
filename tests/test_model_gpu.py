"""GPU end-to-end parity of the drop-in MMVit4 (HIP kernels) against (a) the CPU oracle run on the spot and
(b) the committed fixtures captured from the upstream reference.  Tolerances follow SURVEY section 7 H1:
tight with O(1) activations ('tame' weights), bracketed by the reference's own fp32-vs-fp64 gap for kaiming weights."""
import json
import os

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
META = json.load(open(os.path.join(helpers.GOLDEN, "meta.json")))
CASES = {c["name"]: c for c in META["cases"]}


def sample(t, n=64):
    f = t.detach().reshape(-1)
    n = min(n, f.numel())
    idx = (torch.arange(n, dtype=torch.int64, device=f.device) * (f.numel() - 1)) // max(n - 1, 1)
    return f[idx].double().cpu().numpy()


def build_hip(case):
    import mmvit4
    model = mmvit4.MMVit4()
    sd = helpers.make_state_dict(model.state_dict(), seed=case["wseed"], conv_gain=case["conv_gain"])
    model.load_state_dict(sd)
    model = model.to(DEV)
    if case["mode"] == "eval":
        model.eval()
    else:
        model.train()
        for m in model.modules():                       # train-nodrop: batch statistics, dropout off
            if isinstance(getattr(m, "p", None), float):
                m.p = 0.0
    return model, sd


def run_hip(case):
    import ops
    model, sd = build_hip(case)
    x, mask = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
    pred = model(x.to(DEV))
    loss = None
    if case["mode"] != "eval":
        loss = ops.bce_with_logits_mean(pred, mask.to(DEV))
        loss.backward()
    torch.cuda.synchronize()
    return model, pred, mask, loss, sd


# tame_eval_b3_d3_224: B | 3 at 224^2 (the other branch of the inter-modal re-view, mmvit4.py:481-491).  BASELINE configs[0]
# (tame_train_b4_d4_224) is checked by test_large_baseline_configs_fwd_bwd: its fp64 reference run does not fit the development container.
@pytest.mark.parametrize("name", ["tame_train_b2_d3_64", "tame_eval_b3_d3_64", "tame_train_b1_d4_224", "tame_eval_b3_d3_224"])
def test_against_reference_fixture_tame(name):
    import mmvit4
    case = CASES[name]
    g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
    model, pred, mask, loss, _ = run_hip(case)
    ps = pred.detach()[:, :, 0, ::4, ::4].double().cpu().numpy()
    # Bracketed fp32 tolerance: 16 BatchNorm'd bottlenecks amplify fp32 rounding ~50x, so the
    # reference's OWN fp32 run differs from its fp64 run by `gap` (1.7e-4 train / 2e-5 eval on these fixtures).
    # The HIP path must stay within 3x that gap of the fp64 reference (floor 2e-5 on the sigmoid output).
    gap = np.abs(g["f32/pred_sample"] - g["f64/pred_sample"]).max()
    assert np.abs(ps - g["f64/pred_sample"]).max() < max(3 * gap, 2e-5), (np.abs(ps - g["f64/pred_sample"]).max(), gap)
    n = case["B"] * 224 * 224
    j = mmvit4.Jaccard2(mask[:, 0].reshape(n, 1).to(DEV), pred.detach()[:, 0].reshape(n, 1)).cpu().numpy()
    assert abs(float(j[0]) - float(g["f64/jaccard2"][0])) < 1e-5        # BASELINE: Jaccard within 1e-5 of the reference
    if loss is not None:
        assert abs(loss.item() - float(g["f64/loss"])) < max(1e-5, 3 * abs(float(g["f32/loss"]) - float(g["f64/loss"])))
        params = dict(model.named_parameters())
        worst = 0.0
        for k in helpers.GRAD_KEYS:
            ref = g["f64/grad_sample/" + k]
            got = sample(params[k].grad)
            scale = max(np.abs(ref).max(), 1e-12)
            err = np.abs(got - ref).max() / scale
            ref32 = np.abs(g["f32/grad_sample/" + k] - ref).max() / scale     # the reference's own fp32 error
            worst = max(worst, err)
            assert err < max(1e-3, 10 * ref32), (k, err, ref32)
            nr = float(g["f64/grad_norm/" + k])
            n32 = abs(float(g["f32/grad_norm/" + k]) - nr)                    # the reference's own fp32 error on this norm
            # BatchNorm affine gradients upstream of another BatchNorm are sums that cancel to ~0 in exact arithmetic
            # (d/dbeta of a BN feeding a conv->BN is 0 through that path), so their relative fp32 noise is large: 1e-2.
            assert abs(params[k].grad.double().norm().item() - nr) < max(1e-2 * nr, 10 * n32) + 1e-9, k
        nog = [k for k, p in params.items() if p.grad is None]
        assert len(nog) == 18 and all(k.startswith(helpers.NOGRAD_PREFIXES) for k in nog)
        sd = model.state_dict()
        for k in ("RGB_encoder.e1_bn.running_mean", "RGB_encoder.e1_bn.running_var", "SWIR_encoder.e5.2.bn3.running_var",
                  "NIR_encoder.e3.0.downsample.1.running_mean"):
            np.testing.assert_allclose(sample(sd[k]), g["f64/buf/" + k], rtol=1e-4, atol=1e-6)
        assert int(sd["RGB_encoder.e1_bn.num_batches_tracked"]) == 1


@pytest.mark.parametrize("name", ["kaiming_train_b2_d3_96", "kaiming_train_b2_d4_224"])
def test_against_reference_fixture_kaiming_bracketed(name):
    """The reference's REAL init scale (kaiming_normal_, mmvit4.py:437-439), also at the BASELINE size 224^2 with B = 2 (non-trivial
    inter-modal re-view and BatchNorm batch statistics).  Activations reach 1e3-1e4 and the 3-way correlation softmax saturates, so
    the bar is k x the reference's own fp32-vs-fp64 error: k = 5 for the prediction and the loss, k = 10 (as for the tame
    fixtures) for every sampled gradient and gradient norm of helpers.GRAD_KEYS (45 tensors covering every kernel family); soft
    Jaccard stays within 1e-5.  A gradient norm is not asked to be more accurate than the tensor's sampled elements: for sums that
    cancel analytically (e1_bn.bias: every path but adapt1's is annihilated by the next BatchNorm) the reference's own norm error is
    small only by accident (8e-5 next to 1e-2 on its elements)."""
    import mmvit4
    case = CASES[name]
    g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
    model, pred, mask, loss, _ = run_hip(case)
    ps = pred.detach()[:, :, 0, ::4, ::4].double().cpu().numpy()
    ref_gap = np.abs(g["f32/pred_sample"] - g["f64/pred_sample"]).max()
    assert np.abs(ps - g["f64/pred_sample"]).max() < max(5 * ref_gap, 1e-4)
    n = case["B"] * 224 * 224
    j = mmvit4.Jaccard2(mask[:, 0].reshape(n, 1).to(DEV), pred.detach()[:, 0].reshape(n, 1)).cpu().numpy()
    assert abs(float(j[0]) - float(g["f64/jaccard2"][0])) < 1e-5
    assert abs(loss.item() - float(g["f64/loss"])) < max(1e-4, 5 * abs(float(g["f32/loss"]) - float(g["f64/loss"])))
    params = dict(model.named_parameters())
    bad = []
    for k in helpers.GRAD_KEYS:
        ref = g["f64/grad_sample/" + k]
        got = sample(params[k].grad)
        scale = max(np.abs(ref).max(), 1e-30)
        err = np.abs(got - ref).max() / scale
        ref32 = np.abs(g["f32/grad_sample/" + k] - ref).max() / scale            # the reference's own fp32 error on these samples
        nr = float(g["f64/grad_norm/" + k])
        nerr = abs(params[k].grad.double().norm().item() - nr) / max(nr, 1e-30)
        n32 = abs(float(g["f32/grad_norm/" + k]) - nr) / max(nr, 1e-30)
        if err > max(10 * ref32, 2e-3) or nerr > max(10 * n32, 10 * ref32, 1e-2):      # norm floor 1e-2 as for the tame fixtures

            bad.append((k, err, ref32, nerr, n32))
    assert not bad, bad
    nog = [k for k, p in params.items() if p.grad is None]
    assert len(nog) == 18 and all(k.startswith(helpers.NOGRAD_PREFIXES) for k in nog)


def _ncdhw_sample(t, n=256):
    """sample() of the golden generator on the reference's NCDHW layout, for a channels-last [B, D, H, W, C] activation (tokens
    [B, N, C] are laid out identically in both)"""
    if t.dim() == 5:
        t = t.permute(0, 4, 1, 2, 3).contiguous()
    return sample(t, n)


@pytest.mark.parametrize("name", ["tame_train_b2_d3_64", "tame_eval_b3_d3_224", "kaiming_train_b2_d4_224"])
def test_stage_taps_against_reference_fixture(name):
    """Per-stage parity: nine intermediate activations (first encoder layer, last encoder layer, an early-fusion block, an
    intra-modality transformer, the correlation's qkv conv, the multimodal transformer's input and output, x6_inter, a decoder
    stage at 16^3 and the last one at 128^3) captured from the reference with forward hooks, so a regression is located at its
    stage and not only seen at the sigmoid output.  Bar per tap: max error relative to the tap's largest value <= 5 x the
    reference's own fp32-vs-fp64 figure (floor 2e-5; kaiming: 1e-3)."""
    case = CASES[name]
    g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
    model, _ = build_hip(case)
    model.decoder_split = 0                      # one lane: module hooks then see whole-batch tensors like the reference's
    taps = {}

    def grab(key):
        def hook(mod, inp, out):
            taps[key] = _ncdhw_sample(out)
        return hook

    def pre(mod, inp):
        taps["mm_in"] = _ncdhw_sample(inp[0])

    d = model.decoder_fuse
    for key, mod in (("tap_RGB_e2", model.RGB_encoder.e2), ("tap_SWIR_e5", model.SWIR_encoder.e5), ("tap_fusion3", model.fusion3),
                     ("tap_NIR_transformer", model.NIR_transformer), ("tap_qkv_RGB", model.qkv_RGB),
                     ("tap_mm_transformer", model.multimodal_transformer), ("tap_x6_inter", model.multimodal_decode_conv),
                     ("tap_d4_c2", d.d4_c2), ("tap_d1_out", d.d1_out)):
        mod.register_forward_hook(grab(key))
    model.multimodal_transformer.register_forward_pre_hook(pre)
    x, _ = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
    with torch.no_grad():
        model(x.to(DEV))
    torch.cuda.synchronize()
    floor = 1e-3 if name.startswith("kaiming") else 2e-5
    bad = []
    for key in sorted(taps):
        r64, r32 = g["f64/" + key], g["f32/" + key]
        scale = max(np.abs(r64).max(), 1e-30)
        err = np.abs(taps[key] - r64).max() / scale
        gap = np.abs(r32 - r64).max() / scale
        if err > max(5 * gap, floor):
            bad.append((key, err, gap))
    assert len(taps) == 10 and not bad, bad


def test_full_gradient_against_oracle():
    """EVERY parameter gradient (not a sample) and every buffer against the oracle on the same inputs.
    Bracketed: the oracle is run in fp64 (truth) and in fp32 twice - on the CPU (= the reference's arithmetic, bit-identical to it) and
    on the device (stock ATen kernels, MIOpen off): the same modules, the same fp32, two summation orders.  Per parameter the HIP path's
    rel-L2 error vs fp64 must stay within 4 x the larger of the two fp32 evaluations' own errors (floor 2e-4; 8 x for the named tensors
    below), and the MEDIAN ratio over the 645 tensors within 2.
    What rounds 2-3 measured about this bar: (i) per LAYER, against fp64 on identical inputs, every kernel family is
    at or below ATen's local error (tools/local_error.py; since the GEMM family forms its products from exactly split bf16 terms with
    one fp32 rounding per 16 products, the long-K forward / data-gradient GEMMs went from 1.9-4.5x ATen's error to 0.3-1.6x);
    (ii) end to end a tensor's ratio is governed by WHICH rounding realisation the network's ~50x amplification happens to see: the
    median of this case measured 1.53-1.58 in round 2, 0.94-1.63 across round 3's kernel changes, and over weight seeds 11 / 4 / 7 / 23
    0.97 / 0.60 / 0.76 / 1.34 (profiles/r03_fullgrad_seeds.txt).  A median bar of 1.5 therefore fails or passes by the luck of the
    summation order; 2 with the per-tensor 4x bar is what a kernel that really lost accuracy cannot meet.
    Named 8x tensors: (a) biases of a conv -> ReLU -> InstanceNorm block: the normalisation annihilates their gradient except through the
    ReLU pattern, the analytic value is a small difference of large sums and the fp32 oracle's own error is 1e-2 there; (b) the last
    decoder layer d1_out: the head reads depth slice 0 only, so the gradient entering d1_out's InstanceNorm is zero on 127 of 128 slices
    and what reaches the convolution is rstd (g - mean g - xhat mean(g xhat)): two cancelling scalars per (sample, channel) decide 99 %
    of its elements.  The two fp32 evaluations of the ORACLE differ by 2x on it (CPU 9.9e-5, device 2.0e-4 at seed 11); the HIP path sits
    at 0.2-1.0x of the device oracle's error at seeds 4 / 7 / 23 and at 5x at seed 11 - with its input x closer to fp64 than either
    oracle's (tools/stage_probe.py: 4.0e-5 vs 5.0e-5; the excess is entirely in the incoming gradient).
    64^2 input at batch 2: e4 / e5 still see 4x4 and 2x2 maps (96 / 24 samples per BatchNorm channel); at 32^2 (round 1) e5 normalised
    over 3 samples and both the oracle's fp32 error and ours were O(1) there, which tested nothing."""
    from oracle import mmvit4_oracle as O
    case = dict(B=2, D=3, H=64, W=64, mode="train_nodrop", conv_gain=1.0, wseed=11)
    model, pred, mask, loss, sd = run_hip(case)
    x, _ = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
    res = {}
    for dt in (torch.float32, torch.float64):
        ref = O.MMVit4()
        ref.load_state_dict(sd)
        ref = ref.to(dt).train()
        O.set_dropout(ref, False)
        pr = ref(x.to(dt))
        lr = O.train_step_loss(pr, mask.to(dt))
        lr.backward()
        res[dt] = (pr.detach().double(), lr.item(), {k: (None if p.grad is None else p.grad.double()) for k, p in ref.named_parameters()},
                   {k: b.double() for k, b in ref.named_buffers()})
    p32, l32, g32, b32 = res[torch.float32]
    p64, l64, g64, b64 = res[torch.float64]
    refd = O.MMVit4()                                  # second fp32 evaluation of the oracle: stock ATen on the device
    refd.load_state_dict(sd)
    refd = refd.to(device=DEV).train()
    O.set_dropout(refd, False)
    with torch.backends.cudnn.flags(enabled=False):
        O.train_step_loss(refd(x.to(DEV)), mask.to(DEV)).backward()
    g32d = {k: (None if p.grad is None else p.grad.double().cpu()) for k, p in refd.named_parameters()}
    del refd
    gap = (p32 - p64).abs().max().item()
    assert (pred.detach().cpu().double() - p64).abs().max().item() < max(3 * gap, 2e-5)
    assert abs(loss.item() - l64) < max(3 * abs(l32 - l64), 2e-6)
    bad, ratios = [], []
    for k, p in model.named_parameters():
        if g64[k] is None:
            assert p.grad is None, k
            continue
        nrm = g64[k].norm().clamp_min(1e-20)
        e_hip = ((p.grad.cpu().double() - g64[k]).norm() / nrm).item()
        e_cpu = ((g32[k] - g64[k]).norm() / nrm).item()
        e_ref = max(e_cpu, ((g32d[k] - g64[k]).norm() / nrm).item())
        ratios.append(e_hip / max(e_cpu, 1e-6))
        lim = 8 if k.endswith(".conv.bias") or k.startswith("decoder_fuse.d1_out.conv.") else 4      # named: see the docstring
        if e_hip > max(lim * e_ref, 2e-4):
            bad.append((k, e_hip, e_ref))
    assert not bad, bad[:10]                                   # per tensor: <= 4 x the fp32 oracle's own error (floor 2e-4)
    assert sorted(ratios)[len(ratios) // 2] <= 2.0, sorted(ratios)[len(ratios) // 2]
    for k, b in model.named_buffers():                # running statistics after one training step, bracketed the same way
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(b64[k]), k
            continue
        nrm = b64[k].norm().clamp_min(1e-20)
        e_hip = ((b.cpu().double() - b64[k]).norm() / nrm).item()
        e_ref = ((b32[k] - b64[k]).norm() / nrm).item()
        assert e_hip <= max(4 * e_ref, 5e-5), (k, e_hip, e_ref)


@pytest.mark.parametrize("B,D,HW", [(1, 8, 256), (1, 12, 512), (3, 5, 100)])
def test_other_baseline_configs_forward(B, D, HW):
    """BASELINE configs[2] / [4] geometry (8 bands 256^2, 12 bands 512^2) and an odd size, eval mode, against the CPU oracle.
    Eval-mode fp32 noise of the reference itself is ~2e-5 on the sigmoid output (fixtures); bound 1.5e-4."""
    from oracle import mmvit4_oracle as O
    case = dict(B=B, D=D, H=HW, W=HW, mode="eval", conv_gain=1.0, wseed=31)
    model, sd = build_hip(case)
    x, _ = helpers.make_inputs(B, D, HW, HW)
    with torch.no_grad():
        pg = model(x.to(DEV))
        ref = O.MMVit4()
        ref.load_state_dict(sd)
        ref.eval()
        pr = ref(x)
    assert pg.shape == (B, 3, 1, 224, 224)
    assert (pg.cpu() - pr).abs().max().item() < 1.5e-4


def _oracle_on_device(case, dtype, sd, device=DEV, all_grads=False):
    """The oracle's module tree (stock torch ops, oracle/mmvit4_oracle.py) evaluated ON THE GPU BOX'S DEVICE in `dtype`: forward +
    loss + backward in train-nodrop mode.  Used as the checker where the CPU would need hours (12 bands 512^2: the fp64 CPU run of
    the 8-band case alone takes > 20 min on the box's host share).  test_device_oracle_is_pinned_by_the_reference_fixture ties this
    evaluation to the upstream reference's own fp64 numbers."""
    from oracle import mmvit4_oracle as O
    ref = O.MMVit4()
    ref.load_state_dict(sd)
    ref = ref.to(device=device, dtype=dtype).train()
    O.set_dropout(ref, False)
    x, mask = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
    x, mask = x.to(device=device, dtype=dtype), mask.to(device=device, dtype=dtype)
    # MIOpen off: its per-geometry kernel search / compilation costs minutes per case in fp32 on a fresh box (fp64 never goes through
    # it); ATen's own convolution kernels are slower per call and need no warm-up
    with torch.backends.cudnn.flags(enabled=False):
        pred = ref(x)
        loss = O.train_step_loss(pred, mask)
        loss.backward()
    torch.cuda.synchronize()
    out = {"pred_sample": pred.detach()[:, :, 0, ::4, ::4].double().cpu().numpy(), "loss": loss.item(),
           "grads": {k: p.grad.double() for k, p in ref.named_parameters() if p.grad is not None and (all_grads or k in helpers.GRAD_KEYS)},
           "bufs": {k: ref.state_dict()[k].double().cpu() for k in ("RGB_encoder.e1_bn.running_mean", "SWIR_encoder.e5.2.bn3.running_var")}}
    n = case["B"] * 224 * 224
    out["jaccard2"] = helpers.jaccard2_ref(mask[:, 0].reshape(n, 1), pred.detach()[:, 0].reshape(n, 1)).item()
    del ref, pred, loss
    torch.cuda.empty_cache()
    return out


def test_device_oracle_is_pinned_by_the_reference_fixture():
    """The fp64 device evaluation of the oracle reproduces the upstream reference's own fp64 run (fixture captured by importing the
    reference, tests/golden/make_golden.py) to 5e-9 on the prediction and 1e-7 relative on sampled gradients: it is a valid 'truth'
    for the sizes below."""
    name = "tame_train_b2_d3_64"
    case = CASES[name]
    g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
    from oracle import mmvit4_oracle as O
    sd = helpers.make_state_dict(O.MMVit4().state_dict(), seed=case["wseed"], conv_gain=case["conv_gain"])
    r = _oracle_on_device(case, torch.float64, sd)
    assert np.abs(r["pred_sample"] - g["f64/pred_sample"]).max() < 5e-9        # fp64 on the device vs fp64 on the CPU: 1.0e-9 measured
    assert abs(r["loss"] - float(g["f64/loss"])) < 1e-10
    for k in helpers.GRAD_KEYS:
        ref = g["f64/grad_sample/" + k]
        got = sample(r["grads"][k])
        assert np.abs(got - ref).max() <= 1e-7 * max(np.abs(ref).max(), 1e-30), k


@pytest.mark.parametrize("name,force_single_stream", [("tame_train_b4_d4_224", False), ("oracle_train_b2_d8_256", True),
                                                      ("oracle_train_b2_d12_512", False), ("device_train_b8_d4_224", False)])
def test_large_baseline_configs_fwd_bwd(name, force_single_stream, monkeypatch):
    """BASELINE configs[0] (batch 4, 4 bands, 224^2: the one configuration the reference itself runs, F4_TRAIN.py:52-61; fixture =
    the upstream reference's own fp32 run), configs[2] (8 bands, 256^2) and configs[4] (12 bands, 512^2) geometry, forward AND
    backward, at batch >= 2 (so the inter-modal re-view and the BatchNorm batch statistics are non-trivial).  Truth = the oracle's modules evaluated on the device in
    fp64 (pinned by the test above); the fp32 side of the bracket (the reference's arithmetic) is the committed fixture where one exists
    - the upstream reference's own run for configs[0], the CPU oracle's run on the GPU box's host for the 8-band case
    (tests/golden/make_golden_large.py, 12 min of host time) - and the same modules on the device in fp32 for the 12-band case.
    The 8-band case runs through the single-stream schedule `MMVit4.forward` falls back to near the HBM capacity (what configs[2] at
    B = 64 and configs[4] at B = 16 actually execute), forced here by patching the memory estimate.  Brackets as for the reference
    fixtures: 3x the fp32 arithmetic's own error against fp64 for the prediction, 10x (floor 1e-3) for the sampled gradients."""
    import mmvit4
    if name in helpers.LARGE_CASES or name in helpers.DEVICE_CASES:
        B, D, H, W, wseed = (helpers.LARGE_CASES.get(name) or helpers.DEVICE_CASES[name])
        case = dict(B=B, D=D, H=H, W=W, mode="train_nodrop", conv_gain=1.0, wseed=wseed)
    else:
        case = CASES[name]
        B = case["B"]
    if force_single_stream:
        monkeypatch.setattr(mmvit4, "_memory_limited", lambda x, frac=0.6: True)
    model, pred, mask, loss, sd = run_hip(case)
    assert model.concurrent_branches and model.decoder_split == 2          # the fall-back restores the switches
    ps = pred.detach()[:, :, 0, ::4, ::4].double().cpu().numpy()
    n = B * 224 * 224
    jac = mmvit4.Jaccard2(mask[:, 0].reshape(n, 1).to(DEV), pred.detach()[:, 0].reshape(n, 1)).cpu().numpy()
    hip_loss = loss.item()
    grads = {k: p.grad.double().cpu() for k, p in model.named_parameters() if k in helpers.GRAD_KEYS}
    bufs = {k: v.double().cpu() for k, v in model.state_dict().items() if k in ("RGB_encoder.e1_bn.running_mean", "SWIR_encoder.e5.2.bn3.running_var")}
    assert sum(1 for p in model.parameters() if p.grad is None) == 18
    del model, pred, loss
    torch.cuda.empty_cache()
    r64 = _oracle_on_device(case, torch.float64, sd)
    path = os.path.join(helpers.GOLDEN, name + ".npz")
    if os.path.exists(path):
        # fp32 side of the bracket = the committed fixture: the upstream reference's own fp32 run (configs[0]) / the CPU oracle's fp32 run on
        # the GPU box's host (8 bands).  (Evaluating the oracle on the device in fp32 as well costs minutes of MIOpen kernel search per
        # geometry and adds nothing the fixture does not already pin.)
        g = np.load(path)
        gap = np.abs(g["f32/pred_sample"] - r64["pred_sample"]).max()
        assert gap < 2e-3, gap                                   # the fixture and the device truth describe the same run
        assert np.abs(ps - r64["pred_sample"]).max() < max(3 * gap, 5e-5), (np.abs(ps - r64["pred_sample"]).max(), gap)
        assert abs(float(jac[0]) - r64["jaccard2"]) < 1e-5
        assert abs(hip_loss - r64["loss"]) < max(1e-5, 3 * abs(float(g["f32/loss"]) - r64["loss"]))
        assert np.abs(ps - g["f32/pred_sample"]).max() < max(4 * gap, 5e-5)
        bad = []
        for k in helpers.GRAD_KEYS:              # sampled gradients: HIP and the fixture against the fp64 truth
            t = sample(r64["grads"][k])
            scale = max(np.abs(t).max(), 1e-30)
            e_fix = np.abs(g["f32/grad_sample/" + k] - t).max() / scale
            e_hip = np.abs(sample(grads[k]) - t).max() / scale
            if e_hip > max(10 * e_fix, 1e-3):
                bad.append((k, e_hip, e_fix))
        assert not bad, bad
        for k, b in bufs.items():
            t = r64["bufs"][k]
            assert ((b - t).norm() / t.norm()).item() < 1e-4, k
        return
    r32 = _oracle_on_device(case, torch.float32, sd)
    r32g = {k: v.cpu() for k, v in r32["grads"].items()}
    r32["grads"] = None
    gap = np.abs(r32["pred_sample"] - r64["pred_sample"]).max()
    assert np.abs(ps - r64["pred_sample"]).max() < max(3 * gap, 5e-5), (np.abs(ps - r64["pred_sample"]).max(), gap)
    assert abs(float(jac[0]) - r64["jaccard2"]) < 1e-5
    assert abs(hip_loss - r64["loss"]) < max(1e-5, 3 * abs(r32["loss"] - r64["loss"]))
    bad = []
    for k in helpers.GRAD_KEYS:
        t = r64["grads"][k].cpu()
        nrm = t.norm().clamp_min(1e-30)
        e_hip = ((grads[k] - t).norm() / nrm).item()
        e_ref = ((r32g[k] - t).norm() / nrm).item()
        if e_hip > max(10 * e_ref, 1e-3):
            bad.append((k, e_hip, e_ref))
    assert not bad, bad
    for k, b in bufs.items():
        t = r64["bufs"][k]
        assert ((b - t).norm() / t.norm()).item() < max(3 * ((r32["bufs"][k] - t).norm() / t.norm()).item(), 1e-4), k


def test_headline_batch_taps_up_to_x6_inter():
    """BASELINE configs[1] at ITS OWN batch (B = 32, 4 bands, 224^2; the configuration bench.py times): every stage up to x6_inter -
    encoders (BatchNorm statistics over 32 samples), early fusion, the intra-modality transformers, the inter-modal correlation with
    its batch-dependent re-view (mmvit4.py:481-487), the 2048-token multimodal transformer and multimodal_decode_conv - against the
    oracle's modules evaluated on the device in fp64 (truth) and fp32 (bracket), MIOpen off.  The decoder is left out on the oracle's
    side only because stock ATen cannot run it at this batch ("input tensor must fit into 32-bit index math" at 32 x 32 x 128^3); the
    decoder has no cross-sample coupling and is checked at B = 8 / 4 / 2 by the cases above.  Bar per tap as for the stage taps of the
    reference fixtures: 5 x the fp32 arithmetic's own error against fp64, floor 2e-5 of the tap's largest value."""
    from oracle import mmvit4_oracle as O
    import torch.nn as nn
    case = dict(B=32, D=4, H=224, W=224, mode="train_nodrop", conv_gain=1.0, wseed=44)
    model, sd = build_hip(case)
    model.decoder_split = 0
    taps = {}

    def grab(store, key, conv=None):
        def hook(mod, inp, out):
            store[key] = (conv or sample)(out, 512)
        return hook

    names = (("e2", "RGB_encoder.e2"), ("e5", "SWIR_encoder.e5"), ("fusion3", "fusion3"), ("fusion6", "fusion6"),
             ("NIR_transformer", "NIR_transformer"), ("qkv_RGB", "qkv_RGB"), ("qkv_SWIR", "qkv_SWIR"),
             ("mm_transformer", "multimodal_transformer"), ("x6_inter", "multimodal_decode_conv"))
    mods = dict(model.named_modules())
    for key, mn in names:
        mods[mn].register_forward_hook(grab(taps, key, _ncdhw_sample))
    model.multimodal_transformer.register_forward_pre_hook(lambda mod, inp: taps.__setitem__("mm_in", _ncdhw_sample(inp[0], 512)))
    x, _ = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
    with torch.no_grad():
        model(x.to(DEV))
    torch.cuda.synchronize()
    del model
    torch.cuda.empty_cache()

    class _Stop(Exception):
        pass

    class _NoDecoder(nn.Module):
        def forward(self, *a):
            raise _Stop()

    ref_taps = {}
    for dt in (torch.float64, torch.float32):
        ref = O.MMVit4()
        ref.load_state_dict(sd)
        ref.decoder_fuse = _NoDecoder()
        ref = ref.to(device=DEV, dtype=dt).train()
        O.set_dropout(ref, False)
        store = ref_taps[dt] = {}
        rmods = dict(ref.named_modules())
        for key, mn in names:
            rmods[mn].register_forward_hook(grab(store, key))
        ref.multimodal_transformer.register_forward_pre_hook(lambda mod, inp, store=store: store.__setitem__("mm_in", sample(inp[0], 512)))
        with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
            try:
                ref(x.to(device=DEV, dtype=dt))
            except _Stop:
                pass
        torch.cuda.synchronize()
        del ref
        torch.cuda.empty_cache()
    bad = []
    for key in sorted(taps):
        r64, r32 = ref_taps[torch.float64][key], ref_taps[torch.float32][key]
        scale = max(np.abs(r64).max(), 1e-30)
        err = np.abs(taps[key] - r64).max() / scale
        gap = np.abs(r32 - r64).max() / scale
        if err > max(5 * gap, 2e-5):
            bad.append((key, err, gap))
    assert len(taps) == 10 and not bad, bad


def test_module_surface():
    """the calls F2_MAIN / F4_TRAIN make on the model."""
    import mmvit4
    import torch.nn as nn
    model = mmvit4.MMVit4()
    assert not any(isinstance(m, nn.Conv2d) for m in model.modules())      # init_weights() of F2_MAIN.py:134-157 is a no-op
    assert isinstance(str(model), str) and len(str(model)) > 1000          # F2_MAIN.py:282
    inv = json.load(open(os.path.join(helpers.GOLDEN, "state_dict_inventory.json")))
    sd = model.state_dict()
    assert list(sd.keys()) == list(inv.keys())
    assert all(list(sd[k].shape) == inv[k][0] for k in inv)
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 3, 32, 32))                                 # no CPU fall-back
    model = model.to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)                     # F2_MAIN.py:168-169
    x, mask = helpers.make_inputs(1, 3, 32, 32)
    model.train()
    opt.zero_grad()
    out = model(x.to(DEV))
    loss = torch.nn.BCEWithLogitsLoss()(out, mask.to(DEV))                  # the harness' own loss object (F4_TRAIN.py:58-60)
    loss.backward()
    opt.step()
    assert out.shape == (1, 3, 1, 224, 224) and 0 < out.min().item() and out.max().item() < 1
    assert torch.isfinite(loss).item()
    model.eval()
    with torch.no_grad():
        o2 = model(x.to(DEV))
    assert torch.isfinite(o2).all()
    del model
    torch.cuda.empty_cache()                                                # F2_MAIN.py:307-308


def test_multi_consumer_gradients_ride_in_gemm_epilogues():
    """Every Bottleneck input (residual add or downsample conv next to conv1) and every layer input (adapt conv as a third consumer)
    hands its extra gradients to conv1's data-gradient GEMM epilogue (ops.grad_tap): 3 encoders x (16 blocks + 4 adapt taps) = 60
    tapped gradients per step, none of them late (fallen back to autograd's accumulation) and none needing an add pass of its own;
    switching the mechanism off gives the same gradients up to the association of the sums."""
    import mmvit4
    import ops
    case = dict(B=2, D=3, H=64, W=64, mode="train_nodrop", conv_gain=1.0, wseed=3)
    for k in ops.TAP_STATS:
        ops.TAP_STATS[k] = 0
    for k in ops.NORM_BWD_STATS:
        ops.NORM_BWD_STATS[k] = 0
    m1, _, _, _, _ = run_hip(case)
    assert ops.TAP_STATS == {"epilogue": 60, "added": 0, "late": 0}, ops.TAP_STATS
    # ... and the BatchNorm backward reductions ride in the data-gradient epilogue of the consuming convolution: per encoder 13 bn1
    # (not the three stride-2 blocks, whose conv2 gradient is a parity-class multi-launch), 16 bn2, 15 bn3 (every block with a successor);
    # e1_bn, the downsample norms, those three bn1 and e5's last bn3 keep their own reduction pass (159 BatchNorms in all)
    assert ops.NORM_BWD_STATS == {"epilogue": 3 * 44, "pass": 3 * 9}, ops.NORM_BWD_STATS
    mmvit4.GRAD_TAP = False
    try:
        m2, _, _, _, _ = run_hip(case)
    finally:
        mmvit4.GRAD_TAP = True
    g2 = dict(m2.named_parameters())
    for k, p in m1.named_parameters():
        if p.grad is not None:
            assert ((p.grad - g2[k].grad).norm() / g2[k].grad.norm().clamp_min(1e-30)).item() < 2e-2, k      # a lost gradient would be O(1)


@pytest.mark.parametrize("D", [3, 4, 12])
def test_compact_skip_branch_equals_materialised(D):
    """Decoder_fuse evaluates the nearest-up-sampled skip channels' share of d*_c2 on a compact depth grid (three depth classes per
    up-sampling block) and broadcasts it; with the switch off it materialises the up-sampled tensor and the concat buffer as the
    reference does (mmvit4.py:271-287).  Same arithmetic up to the order of the channel sum: prediction and every gradient agree to
    fp32 rounding amplified by the InstanceNorm chain.  4 bands: levels 1-3 (f = 32, 16, 8) take the compact path; 3 bands (the
    reference-native depth, F8_IMAGES4.py:87) and 12 bands (BASELINE configs[4]) do not divide the 16^3..128^3 grids: blocks of
    floor / ceil(n / D) slices (levels 1-3 at 3 bands: 42.7 / 21.3 / 10.7 slices per block; levels 1 at 12 bands: 10.7)."""
    case = dict(B=2, D=D, H=64, W=64, mode="train_nodrop", conv_gain=1.0, wseed=9)
    import ops
    res = []
    for compact in (True, False):
        model, sd = build_hip(case)
        model.decoder_fuse.compact_skips = compact
        x, mask = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
        pred = model(x.to(DEV))
        ops.bce_with_logits_mean(pred, mask.to(DEV)).backward()
        torch.cuda.synchronize()
        res.append((pred.detach(), {k: p.grad.double() for k, p in model.named_parameters() if p.grad is not None}))
        del model, pred
    (p1, g1), (p2, g2) = res
    assert (p1 - p2).abs().max().item() < 2e-5
    assert g1.keys() == g2.keys()
    # The two paths are two fp32 realisations of the same arithmetic (the channel sum of d*_c2 associated differently), so per tensor
    # they may differ by about the sum of their own rounding errors - which depends on the tensor's conditioning (1e-6 for the decoder
    # head, 1e-2 for biases in front of a normalisation).  The bar is therefore tied, tensor by tensor, to the materialised path's OWN
    # error against the fp64 truth (the oracle's modules on the device in fp64): |compact - materialised| <= 4 x that error + 1e-6.
    truth = _oracle_on_device(case, torch.float64, sd, all_grads=True)["grads"]
    bad = []
    for k in g1:
        nrm = truth[k].norm().clamp_min(1e-30)
        diff = ((g1[k] - g2[k]).norm() / nrm).item()
        own = ((g2[k] - truth[k]).norm() / nrm).item()
        if diff > 4 * own + 1e-6:
            bad.append((k, diff, own))
    assert not bad, bad[:10]


def test_grouped_encoders_equal_per_modality_encoders():
    """MMVit4.grouped_encoders: the three modality encoders as one stacked pass (one grouped launch per twin layer, Z = 3) against one
    Encoder.forward per modality on three streams.  Same kernels on the same tiles: the prediction, the loss and the BatchNorm running
    statistics are bit-identical; the data gradients are the same launches too, so parameter gradients can only differ by the row-split
    order of the weight-gradient reductions (a few ulp, no network amplification)."""
    import ops
    case = dict(B=3, D=3, H=64, W=64, mode="train_nodrop", conv_gain=1.0, wseed=13)
    res = []
    for grouped in (True, False):
        model, _ = build_hip(case)
        model.grouped_encoders = grouped
        x, mask = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
        pred = model(x.to(DEV))
        loss = ops.bce_with_logits_mean(pred, mask.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        res.append((pred.detach(), loss.item(), {k: p.grad for k, p in model.named_parameters() if p.grad is not None},
                    {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}))
    (p1, l1, g1, b1), (p2, l2, g2, b2) = res
    assert torch.equal(p1, p2) and l1 == l2
    assert all(torch.equal(b1[k], b2[k]) for k in b1)
    assert g1.keys() == g2.keys() and len(g1) == 645
    worst = max(((g1[k] - g2[k]).norm() / g2[k].norm().clamp_min(1e-30)).item() for k in g1)
    assert worst < 1e-5, worst


def test_determinism():
    case = dict(B=2, D=3, H=32, W=32, mode="train_nodrop", conv_gain=1.0, wseed=3)
    m1, p1, _, l1, _ = run_hip(case)
    m2, p2, _, l2, _ = run_hip(case)
    assert torch.equal(p1, p2) and l1.item() == l2.item()                   # no atomics anywhere: bit-identical reruns
    g2 = dict(m2.named_parameters())
    for k, p in m1.named_parameters():                                      # ... of every gradient too (two live models, both
        if p.grad is not None:                                              # on their own branch / side streams: this also
            assert torch.equal(p.grad, g2[k].grad), k                       # guards the cross-stream allocator hazards)
    # and the concurrent schedule equals the serial one bit for bit
    import mmvit4
    m3, _ = build_hip(case)
    m3.concurrent_branches = False
    m3.decoder_fuse.concurrent_skips = False
    import ops
    x, mask = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
    p3 = m3(x.to(DEV))
    ops.bce_with_logits_mean(p3, mask.to(DEV)).backward()
    torch.cuda.synchronize()
    assert torch.equal(p1, p3)
    g3 = dict(m3.named_parameters())
    for k, p in m1.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, g3[k].grad), k


def test_dropout_train_mode_runs_and_is_unbiased():
    import mmvit4
    import ops
    case = dict(B=1, D=3, H=32, W=32, mode="eval", conv_gain=1.0, wseed=4)
    model, _ = build_hip(case)
    x, mask = helpers.make_inputs(1, 3, 32, 32)
    with torch.no_grad():
        ref = model(x.to(DEV))
        model.train()
        for m in model.modules():            # keep BN in eval so only dropout differs
            if isinstance(m, mmvit4.BatchNorm3dP):
                m.eval()
        ops.manual_seed(1)
        outs = torch.stack([model(x.to(DEV)) for _ in range(8)])
    assert not torch.equal(outs[0], outs[1])
    assert (outs.mean(0) - ref).abs().mean().item() < 0.05
