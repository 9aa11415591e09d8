"""Where does rounding error ENTER?  Per-layer LOCAL error of the HIP kernels next to the reference's arithmetic (GPU box).

    python tools/local_error.py [B D HW wseed]

The end-to-end gradient test only shows that the HIP path's gradients sit ~1.2-1.6x further from fp64 than the fp32 oracle's; because
the network amplifies any upstream rounding ~50x, that number cannot say WHICH kernel adds the extra noise.  This tool can: the oracle's
modules are evaluated once on the device in fp64 with hooks that capture, for every Conv3d / BatchNorm3d / Linear / LayerNorm call, its
input, output, incoming gradient, input gradient and parameter gradients.  Each layer is then recomputed IN ISOLATION from the SAME
(fp32-rounded) inputs twice - by stock ATen in fp32 on the CPU (= the reference's arithmetic) and by the HIP kernel - and both are
compared with the fp64 capture.  Reported: per layer family the median and worst ratio (HIP local error / ATen local error) for the
output, the data gradient and the weight gradient.  A family whose ratio is ~1 adds no more noise than the reference does.
"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import ops  # noqa: E402
from oracle import mmvit4_oracle as O  # noqa: E402

DEV = torch.device("cuda:0")


def rel(a, t):
    return ((a.double().cpu() - t.double().cpu()).norm() / t.double().cpu().norm().clamp_min(1e-300)).item()


def cl(t):          # NCDHW -> channels-last contiguous
    return t.permute(0, 2, 3, 4, 1).contiguous() if t.dim() == 5 else t.contiguous()


def uncl(t):
    return t.permute(0, 4, 1, 2, 3) if t.dim() == 5 else t


def family(name, mod):
    if isinstance(mod, nn.Conv3d):
        k = tuple(mod.kernel_size)
        K = mod.in_channels * k[0] * k[1] * k[2]
        where = "encoder" if "_encoder." in name else ("decoder" if name.startswith("decoder_fuse") else "other")
        kk = "1x1" if k == (1, 1, 1) else ("%dx%dx%d" % k)
        return "conv %s %s K<=%d" % (where, kk, 1 << max(K - 1, 1).bit_length())
    return type(mod).__name__


def main():
    B, D, HW, wseed = (int(v) for v in (sys.argv[1:5] + ["2", "3", "64", "11"][len(sys.argv) - 1:]))
    ops.SPLIT_BF16 = os.environ.get("CORRIF_SPLIT_BF16", "1") == "1"      # 0 = the fp32-input MFMA chain in every GEMM
    print("ops.SPLIT_BF16 =", ops.SPLIT_BF16)
    torch.manual_seed(0)
    ref = O.MMVit4()
    sd = helpers.make_state_dict(ref.state_dict(), seed=wseed, conv_gain=1.0)
    ref.load_state_dict(sd)
    ref = ref.to(device=DEV, dtype=torch.float64).train()
    O.set_dropout(ref, False)
    cap = {}

    def hook(name):
        def fn(mod, inp, out):
            x = inp[0]
            c = cap[name] = {"x": x.detach(), "y": out.detach()}
            out.register_hook(lambda g, c=c: c.__setitem__("gy", g.detach()))
            if x.requires_grad:
                x.register_hook(lambda g, c=c: c.__setitem__("gx_total", g.detach()))
        return fn

    kinds = (nn.Conv3d, nn.BatchNorm3d, nn.Linear, nn.LayerNorm)
    mods = {n: m for n, m in ref.named_modules() if isinstance(m, kinds)}
    for n, m in mods.items():
        m.register_forward_hook(hook(n))
    x, mask = helpers.make_inputs(B, D, HW, HW)
    with torch.backends.cudnn.flags(enabled=False):
        pred = ref(x.to(device=DEV, dtype=torch.float64))
        O.train_step_loss(pred, mask.to(device=DEV, dtype=torch.float64)).backward()
    torch.cuda.synchronize()
    rows = []
    for name, mod in mods.items():
        c = cap.get(name)
        if c is None or "gy" not in c:
            continue
        if isinstance(mod, nn.Conv3d) and (mod.out_channels & 3):
            continue                       # final_conv (8 -> 3) runs inside ops.head, not as a convolution launch
        # fp64 local truth of this layer from the captured input / incoming gradient (the captured x-gradient is the TOTAL over all
        # consumers of x, so the local one is recomputed in fp64 too)
        m64 = copy.deepcopy(mod)
        for p in m64.parameters():
            p.grad = None
        x64 = c["x"].clone().requires_grad_()
        m64.train()
        y64 = m64(x64)
        y64.backward(c["gy"])
        t = {"y": y64.detach(), "gx": x64.grad, "gw": m64.weight.grad}
        # the reference's arithmetic: stock ATen, fp32, CPU, same rounded inputs
        m32 = copy.deepcopy(mod).float().cpu()
        for p in m32.parameters():
            p.grad = None
        x32 = c["x"].float().cpu().requires_grad_()
        g32 = c["gy"].float().cpu()
        y32 = m32(x32)
        y32.backward(g32)
        a = {"y": y32.detach(), "gx": x32.grad, "gw": m32.weight.grad}
        # the HIP kernels, same rounded inputs
        xh = cl(c["x"].float()).requires_grad_()
        gh = cl(c["gy"].float())
        w = mod.weight.detach().float().clone().requires_grad_()
        b = mod.bias.detach().float().clone().requires_grad_() if getattr(mod, "bias", None) is not None else None
        if isinstance(mod, nn.Conv3d):
            if mod.in_channels == 1:           # stem: the kernel reads the NCDHW input of one modality directly
                xs = c["x"].float()[:, 0].contiguous().requires_grad_()
                yh = ops.conv3d(xs, w, b, mod.stride, mod.padding, False)
                xh = xs
            else:
                yh = ops.conv3d(xh, w, b, mod.stride, mod.padding, mod.padding_mode == "replicate" and mod.kernel_size != (1, 1, 1))
        elif isinstance(mod, nn.BatchNorm3d):
            rm, rv = torch.zeros_like(w), torch.ones_like(w)
            yh = ops.batch_norm(xh, w, b, rm, rv, None, False, False, True, 0.1, mod.eps)
        elif isinstance(mod, nn.Linear):
            yh = ops.linear(xh, w, b)
        else:
            yh = ops.layer_norm(xh, w, b)
        yh.backward(gh)
        torch.cuda.synchronize()
        h = {"y": uncl(yh.detach()), "gx": (uncl(xh.grad) if xh.grad is not None and xh.dim() == 5 else xh.grad), "gw": w.grad}
        if isinstance(mod, nn.Conv3d) and mod.in_channels == 1:
            h["gx"], t["gx"], a["gx"] = None, None, None
        row = [family(name, mod), name]
        for key in ("y", "gx", "gw"):
            if t[key] is None or h[key] is None:
                row += [float("nan"), float("nan")]
                continue
            tt = t[key].reshape(h[key].shape) if key != "gx" else t[key]
            row += [rel(h[key].reshape(tt.shape), tt), rel(a[key].reshape(tt.shape), tt)]
        rows.append(row)
        del cap[name], m64, m32, x64, y64, x32, y32, xh, yh, gh
    torch.cuda.empty_cache()
    fams = {}
    for r in rows:
        fams.setdefault(r[0], []).append(r)

    def med(v):
        v = sorted(x for x in v if x == x)
        return v[len(v) // 2] if v else float("nan")

    print("%-34s %4s | %-30s | %-30s | %-30s" % ("family", "n", "output: hip / aten  (ratio med, max)", "data grad", "weight grad"))
    for f in sorted(fams):
        rs = fams[f]
        cells = []
        for o in (2, 4, 6):
            hv, av = [r[o] for r in rs], [r[o + 1] for r in rs]
            ratios = [x / max(y, 1e-12) for x, y in zip(hv, av) if x == x]
            cells.append("%.1e / %.1e (%.2f, %.2f)" % (med(hv), med(av), med(ratios), max(ratios) if ratios else float("nan")))
        print("%-34s %4d | %-30s | %-30s | %-30s" % (f, len(rs), cells[0], cells[1], cells[2]))
    worst = sorted(rows, key=lambda r: -max((r[o] / max(r[o + 1], 1e-12)) for o in (2, 4, 6) if r[o] == r[o]))[:15]
    print("\nworst layers (max ratio over output / data grad / weight grad):")
    for r in worst:
        print("  %-60s y %.1e/%.1e  gx %.1e/%.1e  gw %.1e/%.1e" % tuple([r[1]] + r[2:]))


if __name__ == "__main__":
    main()
