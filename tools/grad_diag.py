"""Gradient-accuracy diagnostic (GPU box): every parameter gradient of the HIP path against the oracle's modules evaluated on the
device in fp64, bracketed by the same modules in fp32, under the schedule / kernel switches of the product.

    python tools/grad_diag.py B D HW [gain] [wseed]

Prints, per variant, how many tensors exceed 10x the fp32 arithmetic's own error (floor 1e-3) and the worst ratios."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import torch  # noqa: E402

import mmvit4  # noqa: E402
import ops  # noqa: E402
from oracle import mmvit4_oracle as O  # noqa: E402

DEV = torch.device("cuda:0")


def oracle(case, dtype, sd):
    ref = O.MMVit4()
    ref.load_state_dict(sd)
    ref = ref.to(device=DEV, dtype=dtype).train()
    O.set_dropout(ref, False)
    x, mask = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
    x, mask = x.to(device=DEV, dtype=dtype), mask.to(device=DEV, dtype=dtype)
    with torch.backends.cudnn.flags(enabled=False):      # no MIOpen kernel search (minutes per geometry in fp32 on a fresh box)
        pred = ref(x)
        O.train_step_loss(pred, mask).backward()
    torch.cuda.synchronize()
    out = {k: p.grad.double().cpu() for k, p in ref.named_parameters() if p.grad is not None}
    out["__pred__"] = pred.detach().double().cpu()
    del ref, pred
    torch.cuda.empty_cache()
    return out


def hip(case, sd, serial, stream_k, tap=True, patch=True, bwd_stats=True, splits_r1=False, stem=True, compact=True, ksplit=True, grouped=True):
    ops.STREAM_K, ops.USE_PATCH, mmvit4.GRAD_TAP = stream_k, patch, tap
    ops.SPLIT_BF16 = ksplit
    ops.BWD_STATS, ops.WGRAD_SPLITS_R1, ops.USE_STEM_KERNEL = bwd_stats, splits_r1, stem
    model = mmvit4.MMVit4()
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    for m in model.modules():
        if isinstance(getattr(m, "p", None), float):
            m.p = 0.0
    model.decoder_fuse.compact_skips = compact
    model.grouped_encoders = grouped
    if serial:
        model.concurrent_branches, model.decoder_split, model.decoder_fuse.concurrent_skips = False, 0, False
    x, mask = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
    pred = model(x.to(DEV))
    ops.bce_with_logits_mean(pred, mask.to(DEV)).backward()
    torch.cuda.synchronize()
    out = {k: p.grad.double().cpu() for k, p in model.named_parameters() if p.grad is not None}
    out["__pred__"] = pred.detach().double().cpu()
    del model, pred
    torch.cuda.empty_cache()
    return out


def main():
    B, D, HW = (int(v) for v in sys.argv[1:4])
    gain = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    wseed = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    case = dict(B=B, D=D, H=HW, W=HW)
    sd = helpers.make_state_dict(O.MMVit4().state_dict(), seed=wseed, conv_gain=gain)
    r64 = oracle(case, torch.float64, sd)
    r32 = oracle(case, torch.float32, sd)
    print("oracle fp32 vs fp64: pred gap %.3e" % (r32["__pred__"] - r64["__pred__"]).abs().max().item(), flush=True)
    variants = [("default", dict(serial=False, stream_k=False)),
                ("fp32-input MFMA chain in every GEMM (rounds 1-3)", dict(serial=False, stream_k=False, ksplit=False)),
                ("one Encoder.forward per modality (twins)", dict(serial=False, stream_k=False, grouped=False)),
                ("BatchNorm backward reductions as their own pass", dict(serial=False, stream_k=False, bwd_stats=False)),
                ("round-1 weight-gradient splits", dict(serial=False, stream_k=False, splits_r1=True)),
                ("both of the above", dict(serial=False, stream_k=False, bwd_stats=False, splits_r1=True)),
                ("scalar-gather stem", dict(serial=False, stream_k=False, stem=False)),
                ("materialised skip branch", dict(serial=False, stream_k=False, compact=False)),
                ("serial, no grad_tap", dict(serial=True, stream_k=False, tap=False)),
                ("stream-K", dict(serial=False, stream_k=True))]
    base = None
    for name, kw in variants:
        h = hip(case, sd, **kw)
        rows = []
        for k, t in r64.items():
            if k == "__pred__":
                continue
            nrm = t.norm().clamp_min(1e-30)
            e_hip = ((h[k] - t).norm() / nrm).item()
            e_ref = ((r32[k] - t).norm() / nrm).item()
            rows.append((e_hip / max(e_ref, 1e-4), e_hip, e_ref, k))
        rows.sort(reverse=True)
        bad = [r for r in rows if r[1] > max(10 * r[2], 1e-3)]
        ratios = sorted(r[0] for r in rows)
        print("\n== %s: pred err %.3e; %d of %d tensors beyond max(10 x fp32 error, 1e-3); ratio median %.2f, p90 %.2f, max %.1f"
              % (name, (h["__pred__"] - r64["__pred__"]).abs().max().item(), len(bad), len(rows), ratios[len(ratios) // 2],
                 ratios[len(ratios) * 9 // 10], ratios[-1]), flush=True)
        for r in rows[:12]:
            print("   %-70s e_hip %.3e  e_ref %.3e  ratio %.1f" % (r[3], r[1], r[2], r[0]))
        if base is None:
            base = h
        else:
            d = max(((h[k] - base[k]).norm() / base[k].norm().clamp_min(1e-30)).item() for k in base if k != "__pred__")
            print("   max rel-L2 difference to the first variant: %.3e" % d)


if __name__ == "__main__":
    main()
