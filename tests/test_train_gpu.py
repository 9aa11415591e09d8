"""GPU: the callers of the hot path: fused Adam, the training step of F4_TRAIN.py:54-71, eval path."""
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _models(seed, train):
    import mmvit4
    from oracle import mmvit4_oracle as O
    ref = O.MMVit4()
    sd = helpers.make_state_dict(ref.state_dict(), seed=seed, conv_gain=1.0)
    ref.load_state_dict(sd)
    hip = mmvit4.MMVit4()
    hip.load_state_dict(sd)
    hip = hip.to(DEV)
    if train:
        ref.train(); O.set_dropout(ref, False)
        hip.train()
        for m in hip.modules():
            if isinstance(getattr(m, "p", None), float):
                m.p = 0.0
    else:
        ref.eval(); hip.eval()
    return ref, hip


def test_fused_adam_matches_torch_adam_on_identical_gradients():
    import train
    torch.manual_seed(0)
    shapes = [(64, 32, 1, 3, 3), (3,), (1000,), (512, 512), (1, 512, 512), (7, 5)]
    ps = [torch.randn(s) for s in shapes]
    ref = [p.clone().requires_grad_() for p in ps]
    hip = [p.clone().to(DEV).requires_grad_() for p in ps]
    extra_ref, extra_hip = torch.randn(5, requires_grad=True), torch.randn(5, device=DEV, requires_grad=True)   # never gets a grad
    o_ref = torch.optim.Adam(ref + [extra_ref], lr=1e-3)
    o_hip = train.FusedAdam(hip + [extra_hip], lr=1e-3)
    before = extra_hip.detach().clone()
    for step in range(4):
        for r, h in zip(ref, hip):
            g = torch.randn(r.shape) * (step + 1)
            r.grad = g.clone()
            h.grad = g.to(DEV)
        o_ref.step()
        o_hip.step()
    torch.cuda.synchronize()
    for r, h in zip(ref, hip):
        assert torch.allclose(h.detach().cpu(), r.detach(), rtol=2e-6, atol=1e-7)
    assert torch.equal(extra_hip.detach(), before)            # grad is None -> skipped, like torch.optim.Adam


def test_train_steps_follow_the_oracle():
    """F4_TRAIN.py:54-71 restated: step 1 must agree to fp32 noise; later steps within the Adam-amplified noise (Adam moves every
    weight by ~lr whatever the gradient magnitude, so sign noise on near-zero gradients perturbs the trajectory at the lr level)."""
    import train
    from oracle import mmvit4_oracle as O
    ref, hip = _models(21, True)
    x, mask = helpers.make_inputs(2, 3, 32, 32)
    o_ref = torch.optim.Adam(ref.parameters(), lr=1e-4)                   # F2_MAIN.py:168-169
    s_ref = torch.optim.lr_scheduler.StepLR(o_ref, step_size=5, gamma=0.9)
    o_hip = train.FusedAdam(hip.parameters(), lr=1e-4)
    s_hip = train.StepLR(o_hip, 5, 0.9)
    s_hip.step()
    l_ref, l_hip, j_hip = [], [], []
    n = 2 * 224 * 224
    for step in range(3):
        o_ref.zero_grad()
        pr = ref(x)
        lr_ = O.train_step_loss(pr, mask)
        lr_.backward()
        o_ref.step()
        l_ref.append(lr_.item())
        loss, jac, nn_ = train.train_step(hip, o_hip, x.to(DEV), mask.to(DEV))
        l_hip.append(loss.item())
        j_hip.append(jac.item() / nn_)
        if step == 0:
            jr = helpers.jaccard2_ref(mask[:, 0].reshape(n, 1), pr.detach()[:, 0].reshape(n, 1)).item()
            assert abs(j_hip[0] - jr) < 1e-5
    assert abs(l_hip[0] - l_ref[0]) < 2e-6
    assert abs(l_hip[1] - l_ref[1]) < 2e-3 and abs(l_hip[2] - l_ref[2]) < 2e-3
    assert o_hip.lr == pytest.approx(1e-4) and s_hip.get_lr() == [o_hip.lr]
    # the 18 grad-less tensors are untouched, everything else moved
    sd0 = helpers.make_state_dict(hip.state_dict(), seed=21, conv_gain=1.0)
    for k, p in hip.named_parameters():
        moved = not torch.equal(p.detach().cpu(), sd0[k])
        assert moved != k.startswith(helpers.NOGRAD_PREFIXES), k


def test_train_trace_against_reference_fixture(tmp_path):
    """N1's pin: three consecutive steps of F4_TRAIN.py:54-71 with the optimiser of F2_MAIN.py:168-173 - the
    upstream model + torch.optim.Adam + StepLR stepped BEFORE the optimiser (F4_TRAIN.py:46; step_size 1 / gamma 0.5 so that the quirk
    shows: the run uses lr/2 from its first step) - captured from the reference in fp32 and fp64 (tests/golden/make_golden.py
    train_trace): per-step loss and soft Jaccard, post-step parameter samples / norms, the sum of |every state-dict entry|, the
    BatchNorm counter, the number of untouched parameters.  The HIP path runs the same three steps through train.train_model
    (FusedAdam, StepLR, the reference's checkpoint names).  Step 1 agrees to fp32 noise; afterwards Adam turns sign noise on
    near-zero gradients into +-lr moves, so steps 2-3 and the final parameters are bracketed by the reference's own fp32-vs-fp64
    difference (x5) with a floor of 2 * steps * lr on single weights."""
    import os
    import numpy as np
    import train
    g = np.load(os.path.join(helpers.GOLDEN, "train_trace_b2_d3_64.npz"))
    _, hip = _models(7, True)
    x, mask = helpers.make_inputs(2, 3, 64, 64)
    optim = train.FusedAdam(hip.parameters(), lr=1e-4)
    sched = train.StepLR(optim, 1, 0.5)

    class Loader:                                    # three batches of one epoch, the same tensors each time
        def __iter__(self):
            return iter([(x, mask)] * 3)

    losses, jacs = [], []
    orig = train.train_step

    def spy(*a, **kw):
        loss, jac, n = orig(*a, **kw)
        losses.append(loss.item())
        jacs.append(jac.item() / n)
        return loss, jac, n

    train.train_step = spy
    try:
        log = train.train_model(1, hip, sched, Loader(), optim, DEV, str(tmp_path), 4)
    finally:
        train.train_step = orig
    assert optim.lr == pytest.approx(float(g["f64/lr"])) == pytest.approx(5e-5)
    l64, l32 = g["f64/loss"], g["f32/loss"]
    assert abs(losses[0] - l64[0]) < max(2e-6, 3 * abs(l32[0] - l64[0]))
    for i in (1, 2):
        assert abs(losses[i] - l64[i]) < max(2e-5, 5 * abs(l32[i] - l64[i])), (i, losses, l64, l32)
        assert abs(jacs[i] - g["f64/jaccard2"][i]) < max(1e-5, 5 * abs(g["f32/jaccard2"][i] - g["f64/jaccard2"][i]))
    assert abs(jacs[0] - g["f64/jaccard2"][0]) < 1e-5
    assert abs(log[0][0] - float(np.mean(l64))) < 1e-4                      # epoch mean of the batch losses (F4_TRAIN.py:73)
    per_epoch, final = train.checkpoint_paths(str(tmp_path), 4)
    assert os.path.basename(per_epoch) == "iremmodel4.pt" and os.path.basename(final) == "Finaliremmodel4.pt"
    sd = torch.load(final, map_location="cpu")
    assert list(sd.keys()) == list(hip.state_dict().keys()) and os.path.exists(per_epoch)
    assert int(sd["RGB_encoder.e1_bn.num_batches_tracked"]) == int(g["f64/nbt"]) == 3
    floor = 2 * 3 * 5e-5
    bad = []
    for k in helpers.GRAD_KEYS + ["RGB_encoder.e1_bn.running_mean", "SWIR_encoder.e5.2.bn3.running_var"]:
        ref, r32 = g["f64/param_sample/" + k], g["f32/param_sample/" + k]
        f = sd[k].reshape(-1)
        idx = (torch.arange(min(64, f.numel()), dtype=torch.int64) * (f.numel() - 1)) // max(min(64, f.numel()) - 1, 1)
        got = f[idx].double().numpy()
        err, e32 = np.abs(got - ref).max(), np.abs(r32 - ref).max()
        nr = float(g["f64/param_norm/" + k])
        nerr, n32 = abs(sd[k].double().norm().item() - nr), abs(float(g["f32/param_norm/" + k]) - nr)
        if err > max(5 * e32, floor) or nerr > max(5 * n32, 1e-4 * nr + 1e-6):
            bad.append((k, err, e32, nerr, n32))
    assert not bad, bad
    tot = sum(v.double().abs().sum().item() for v in sd.values() if v.dtype.is_floating_point)
    t64, t32 = float(g["f64/abs_sum_all"]), float(g["f32/abs_sum_all"])
    assert abs(tot - t64) < max(5 * abs(t32 - t64), 1e-6 * t64)
    frozen = helpers.make_state_dict(sd, seed=7, conv_gain=1.0)
    untouched = sum(1 for k, _ in hip.named_parameters() if torch.equal(sd[k], frozen[k]))
    assert untouched == int(g["f64/untouched"]) == 18


def test_eval_path_and_per_image_metrics():
    import train
    ref, hip = _models(22, False)
    x, mask = helpers.make_inputs(2, 3, 32, 32)
    with torch.no_grad():
        pr = ref(x)
    loss, jac = train.evaluate(hip, [(x, mask)], DEV)
    n = 2 * 224 * 224
    jr = helpers.jaccard2_ref(mask[:, 0].reshape(n, 1), pr[:, 0].reshape(n, 1)).item()
    assert abs(jac - jr) < 1e-5
    assert abs(loss - torch.nn.functional.binary_cross_entropy_with_logits(pr, mask).item()) < 1e-5
    js, fs = train.per_image_metrics(hip, x.to(DEV), mask.to(DEV))
    for i in range(2):
        n1 = 224 * 224
        with torch.no_grad():
            p1 = ref(x[i:i + 1])
        j1 = helpers.jaccard2_ref(mask[i, 0].reshape(n1, 1), p1[0, 0].reshape(n1, 1)).item()
        assert abs(js[i] - j1) < 1e-5 and 0.0 <= fs[i] <= 1.0


def test_reducer_rccl_path_single_rank():
    """the N > 1 code path (bucket views, post-accumulate hooks on three branch streams, async RCCL all-reduce on a side stream)
    rehearsed with ONE rank on the GPU: gradients must equal the plain backward, step after step."""
    import os
    import torch.distributed as dist
    import ops
    from data_parallel import GradAllReducer, broadcast_module_state
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(helpers.free_port()))
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        _, hip = _models(23, True)
        _, hip2 = _models(23, True)
        broadcast_module_state(hip)
        red = GradAllReducer(hip, bucket_bytes=8 << 20, force_collective=True)
        x, mask = helpers.make_inputs(2, 3, 32, 32)
        for step in range(3):
            red.zero_grad()
            ops.bce_with_logits_mean(hip(x.to(DEV)), mask.to(DEV)).backward()
            red.finish()
            for p in hip2.parameters():
                p.grad = None
            ops.bce_with_logits_mean(hip2(x.to(DEV)), mask.to(DEV)).backward()
            torch.cuda.synchronize()
            p2 = dict(hip2.named_parameters())
            for k, p in hip.named_parameters():
                if p2[k].grad is None:
                    assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k
                else:
                    assert torch.equal(p.grad, p2[k].grad), (step, k)
        assert len(red.buckets) > 5 and red.communicated_elements() == 85343883      # the 18 grad-less tensors never travel
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [1, 8])
def test_torch_library_binding_equals_ctypes_binding(B):
    """The eval forward (F4_TRAIN.py:181-208) through torch.ops.corrif.* (descriptors filled in C++, csrc_torch/corrif_torch.cpp) against
    the same forward through the ctypes binding: the same kernels with the same arguments, bit-identical.  B = 1 runs the grouped
    encoder ops (conv3d_grouped_fwd / batch_norm_grouped_eval), B = 8 the per-modality ones."""
    import ops
    _, hip = _models(seed=29, train=False)
    x, _ = helpers.make_inputs(B, 3, 96, 96, seed=5)
    x = x.to(DEV)
    assert ops.USE_TORCH_LIBRARY and ops.tl() is not None
    with torch.no_grad():
        y1 = hip(x).clone()
        was, ops.USE_TORCH_LIBRARY = ops.USE_TORCH_LIBRARY, False
        try:
            y2 = hip(x).clone()
        finally:
            ops.USE_TORCH_LIBRARY = was
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)


def test_hip_graph_eval_forward_is_bit_identical_to_eager():
    """train.GraphedForward: the eval forward captured in a HIP graph WITH the module's multi-stream schedule (branch streams,
    sample-group lanes, decoder skip stream) replays to the eager result, also for a second input written into the static buffer;
    an eager forward and a second (single-stream) capture AFTER it still work - that sequence crashed the process in round 1,
    when fork/join events were destroyed while their stream was capturing (mmvit4._Edges)."""
    import train
    _, hip = _models(seed=21, train=False)
    x1, _ = helpers.make_inputs(1, 3, 64, 64, seed=1)
    x2, _ = helpers.make_inputs(1, 3, 64, 64, seed=2)
    x1, x2 = x1.to(DEV), x2.to(DEV)
    with torch.no_grad():
        e1, e2 = hip(x1).clone(), hip(x2).clone()
    g = train.GraphedForward(hip, x1)
    assert hip.concurrent_branches and hip.decoder_split == 2 and hip.decoder_fuse.concurrent_skips
    assert torch.equal(g(x1), e1)
    assert torch.equal(g(x2), e2)
    assert torch.equal(g(x1), e1)
    with torch.no_grad():
        assert torch.equal(hip(x2), e2)                   # eager forward after the multi-stream capture
    g1 = train.GraphedForward(hip, x2, single_stream=True)
    assert hip.concurrent_branches and hip.decoder_split == 2 and hip.decoder_fuse.concurrent_skips
    assert torch.equal(g1(x1), e1)
    assert torch.equal(g(x2), e2)                         # the first graph is still replayable
    # a second MULTI-stream capture on the same model at another batch size (two sample-group lanes instead of one), with both
    # graphs alive: every captured forward records its own set of fork/join events (tools/graph_eval.py crashed here in round 2
    # while the events of the first graph were re-recorded into the second capture)
    x4, _ = helpers.make_inputs(4, 3, 64, 64, seed=3)
    x4 = x4.to(DEV)
    with torch.no_grad():
        e4 = hip(x4).clone()
    g4 = train.GraphedForward(hip, x4)
    assert torch.equal(g4(x4), e4)
    assert torch.equal(g(x1), e1) and torch.equal(g1(x2), e2)
    with torch.no_grad():
        assert torch.equal(hip(x4), e4)
    torch.cuda.synchronize()


def test_bench_two_rank_control_flow_rehearsal():
    """bench.py under torchrun with 2 ranks on this one GPU (gloo instead of RCCL: same code path, collectives included):
    every rank must reach every collective - a rank-0-only section containing a gradient all-reduce would hang here."""
    import json
    import os
    import subprocess
    import sys
    port = helpers.free_port()
    env = dict(os.environ, CORRIF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--batch", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["scaling"] == "weak"
    assert out["value"] > 0 and "roofline" in out
