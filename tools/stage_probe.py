"""Where does a gradient's end-to-end error enter?  Outputs and output-gradients of the decoder's / fusion modules of the HIP model against
the oracle's modules on the device in fp64 (truth) and fp32 (bracket), with the split-bf16 loop on and off (GPU box).
    python tools/stage_probe.py [B D HW wseed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa
import torch
import ops
import mmvit4
from oracle import mmvit4_oracle as O

DEV = torch.device("cuda:0")
B, D, HW, wseed = (int(v) for v in (sys.argv[1:5] + ["2", "3", "64", "11"][len(sys.argv) - 1:]))
sd = helpers.make_state_dict(O.MMVit4().state_dict(), seed=wseed, conv_gain=1.0)
x, mask = helpers.make_inputs(B, D, HW, HW)


def wanted(n):
    return n.startswith("decoder_fuse.") and n.count(".") == 1 or n in ("multimodal_decode_conv", "fusion4", "fusion5", "fusion6")


def capture(model, store, to_ncdhw):
    def hook(name):
        def fn(mod, inp, out):
            if not torch.is_tensor(out) or out.dim() != 5:
                return
            conv = (lambda t: t.permute(0, 4, 1, 2, 3)) if to_ncdhw else (lambda t: t)
            store[name + " y"] = conv(out.detach()).double().cpu()
            if out.requires_grad:
                out.register_hook(lambda g, name=name: store.__setitem__(name + " gy", conv(g.detach()).double().cpu()))
        return fn
    for n, m in model.named_modules():
        if wanted(n):
            m.register_forward_hook(hook(n))


# the operands of d1_out's weight gradient (8 -> 8 channels, 1x1): the layer input x and the incoming gradient gy, as the HIP path sees them
cap = {}
_bwd = ops.ConvFn.backward


def spy(ctx, gy):
    xs, w = ctx.saved_tensors[:2]
    if tuple(w.shape) == (8, 8, 1, 1, 1):
        print("   [spy] d1_out backward: x", tuple(xs.shape), xs.stride(), " gy", tuple(gy.shape), gy.stride(), flush=True)
        cap.setdefault("d1_out.conv x", []).append(xs.detach().reshape(-1, 8).double().cpu())
        cap.setdefault("d1_out.conv gy", []).append(gy.detach().reshape(-1, 8).double().cpu())
    return _bwd(ctx, gy)


ops.ConvFn.backward = staticmethod(spy)
ocap = {}


def ohook(dt):
    def fn(mod, inp, out):
        ocap[(dt, "d1_out.conv x")] = inp[0].detach().permute(0, 2, 3, 4, 1).reshape(-1, 8).double().cpu()
        out.register_hook(lambda g: ocap.__setitem__((dt, "d1_out.conv gy"), g.detach().permute(0, 2, 3, 4, 1).reshape(-1, 8).double().cpu()))
    return fn


ref = {}
for dt in (torch.float64, torch.float32):
    r = O.MMVit4(); r.load_state_dict(sd); r = r.to(device=DEV, dtype=dt).train(); O.set_dropout(r, False)
    ref[dt] = {}
    capture(r, ref[dt], False)
    dict(r.named_modules())["decoder_fuse.d1_out.conv"].register_forward_hook(ohook(dt))
    with torch.backends.cudnn.flags(enabled=False):
        O.train_step_loss(r(x.to(device=DEV, dtype=dt)), mask.to(device=DEV, dtype=dt)).backward()
    torch.cuda.synchronize()
    for kk in ("d1_out.conv x", "d1_out.conv gy"):
        ref[dt][kk] = ocap[(dt, kk)]
    del r
runs = {}
for split in (True, False):
    ops.SPLIT_BF16 = split
    m = mmvit4.MMVit4(); m.load_state_dict(sd); m = m.to(DEV).train()
    for q in m.modules():
        if isinstance(getattr(q, "p", None), float):
            q.p = 0.0
    runs[split] = {}
    capture(m, runs[split], True)
    ops.bce_with_logits_mean(m(x.to(DEV)), mask.to(DEV)).backward()
    torch.cuda.synchronize()
    for kk, vv in cap.items():
        runs[split][kk] = torch.cat(vv[::-1], 0)          # lanes run their backward in reverse order of the forward
    cap.clear()
    del m
print("%-44s %12s %12s %12s   (rel L2 error vs fp64; ratio = split / fp32-oracle)" % ("stage", "split-bf16", "fp32 MFMA", "oracle fp32"))
for k in sorted(ref[torch.float64], key=lambda s: (s.split(" ")[1], s)):

    t = ref[torch.float64][k]
    if k not in runs[True] or runs[True][k].shape != t.shape:
        continue
    n = t.norm().clamp_min(1e-300)
    e = [((runs[s][k] - t).norm() / n).item() for s in (True, False)] + [((ref[torch.float32][k] - t).norm() / n).item()]
    print("%-44s %12.3e %12.3e %12.3e   %.1f" % (k, e[0], e[1], e[2], e[0] / max(e[2], 1e-30)))

# d1_out's weight gradient recomputed in fp64 from each side's (x, gy): which operand carries the error?
X64, G64 = ref[torch.float64]["d1_out.conv x"], ref[torch.float64]["d1_out.conv gy"]
W64 = G64.t() @ X64
rel = lambda a: ((a - W64).norm() / W64.norm()).item()
print("\nd1_out.conv weight gradient = gy^T x recomputed in fp64 from ... (rel L2 error vs the fp64 oracle's)")
for nm, src in (("split-bf16 run", runs[True]), ("fp32-MFMA run", runs[False]), ("oracle fp32", ref[torch.float32])):
    X, G = src["d1_out.conv x"], src["d1_out.conv gy"]
    print("   %-16s its x and its gy %.3e | its x, true gy %.3e | true x, its gy %.3e" % (nm, rel(G.t() @ X), rel(G64.t() @ X), rel(G.t() @ X64)))

# which REGION's forward arithmetic moves d1_out's incoming gradient?  The split loop switched on per region of the forward pass
# (forward pre-hooks on the top-level children; the backward runs with the decoder's setting)
def region(name):
    if name.endswith("_encoder") or name.startswith("fusion"):
        return "enc"
    return "dec" if name == "decoder_fuse" else "mid"


G64 = ref[torch.float64]["d1_out.conv gy"]
print("\nd1_out.conv gy error vs fp64 with the split loop on in ... (oracle fp32: %.3e)" % ((ref[torch.float32]["d1_out.conv gy"] - G64).norm() / G64.norm()).item())
for on in ({"enc"}, {"mid"}, {"dec"}, {"enc", "mid"}, {"mid", "dec"}, {"enc", "dec"}):
    m = mmvit4.MMVit4(); m.load_state_dict(sd); m = m.to(DEV).train()
    for q in m.modules():
        if isinstance(getattr(q, "p", None), float):
            q.p = 0.0
    for n, c in m.named_children():
        c.register_forward_pre_hook(lambda mod, inp, r=region(n): setattr(ops, "SPLIT_BF16", r in on))
    ops.SPLIT_BF16 = "enc" in on
    ops.bce_with_logits_mean(m(x.to(DEV)), mask.to(DEV)).backward()
    torch.cuda.synchronize()
    g = torch.cat(cap["d1_out.conv gy"][::-1], 0)
    cap.clear()
    print("   %-28s %.3e" % (" + ".join(sorted(on)), ((g - G64).norm() / G64.norm()).item()), flush=True)
    del m
