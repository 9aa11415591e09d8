"""Diagnostic (GPU box, uses the oracle: test infrastructure): where does the MMVit2 decoder-gradient error vs fp64 come from?
Answer: ReLU mask flips at voxels whose conv output is within fp32 noise of zero - see tests/test_mmvit2_gpu.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, mmvit2, ops
from oracle import mmvit2_oracle as O2
from oracle import mmvit4_oracle as O
DEV = "cuda:0"
ops.USE_PATCH = os.environ.get("NO_PATCH") is None
case = dict(B=1, D=4, H=24, W=40, conv_gain=1.0, wseed=12)
model = mmvit2.MMVit2()
sd = helpers.make_state_dict(model.state_dict(), seed=case["wseed"], conv_gain=1.0)
model.load_state_dict(sd); model = model.to(DEV).train()
for m in model.modules():
    if isinstance(getattr(m, "p", None), float): m.p = 0.0
x, mask = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
acts = {}
def hook(name):
    def f(mod, inp, out):
        out.retain_grad(); acts[name] = out
    return f
for n in ("d1_c1", "d1_c2", "d2_c1", "d1_out"):
    getattr(model.decoder_fuse, n).register_forward_hook(hook(n))
    getattr(model.decoder_fuse, n).conv.register_forward_hook(hook(n + ".conv"))
pred = model(x.to(DEV)); loss = ops.bce_with_logits_mean(pred, mask.to(DEV)); loss.backward(); torch.cuda.synchronize()
ref = O2.MMVit2(); ref.load_state_dict(sd); ref = ref.double().train(); O.set_dropout(ref, False)
racts = {}
def rhook(name):
    def f(mod, inp, out):
        out.retain_grad(); racts[name] = out
    return f
for n in ("d1_c1", "d1_c2", "d2_c1", "d1_out"):
    getattr(ref.decoder_fuse, n).register_forward_hook(rhook(n))
    getattr(ref.decoder_fuse, n).conv.register_forward_hook(rhook(n + ".conv"))
pr = ref(x.double()); lr = O.train_step_loss(pr, mask.double()); lr.backward()
def rel(a, b): return ((a - b).norm() / (b.norm() + 1e-30)).item()
ref32 = O2.MMVit2(); ref32.load_state_dict(sd); ref32 = ref32.float().train(); O.set_dropout(ref32, False)
r32 = {}
def r32hook(name):
    def f(mod, inp, out):
        out.retain_grad(); r32[name] = out
    return f
for n in ("d1_c1", "d1_c2", "d2_c1", "d1_out"):
    getattr(ref32.decoder_fuse, n).register_forward_hook(r32hook(n))
    getattr(ref32.decoder_fuse, n).conv.register_forward_hook(r32hook(n + ".conv"))
p32 = ref32(x); l32 = O.train_step_loss(p32, mask); l32.backward()
for n in ("d1_out", "d1_c2", "d1_c1", "d2_c1"):
    print("CPU fp32", n, "act rel", rel(r32[n].detach().double(), racts[n].detach()), "grad rel", rel(r32[n].grad.double(), racts[n].grad),
          "| conv-out grad rel", rel(r32[n + ".conv"].grad.double(), racts[n + ".conv"].grad))
for n in ("d1_out", "d1_c2", "d1_c1", "d2_c1"):
    a = acts[n].detach().cpu().double().permute(0, 4, 1, 2, 3); b = racts[n].detach()
    ga = acts[n].grad.cpu().double().permute(0, 4, 1, 2, 3); gb = racts[n].grad
    print(n, "act rel", rel(a, b), "grad rel", rel(ga, gb), "grad interior rel", rel(ga[..., 2:-2, 2:-2, 2:-2], gb[..., 2:-2, 2:-2, 2:-2]))
    gc = acts[n + ".conv"].grad.cpu().double().permute(0, 4, 1, 2, 3)
    print("   HIP conv-out grad rel", rel(gc, racts[n + ".conv"].grad))
    e = (ga - gb).abs()
    print("   max err at", [int(v) for v in (e == e.max()).nonzero()[0]], "max", e.max().item(), "gb max", gb.abs().max().item())
for k in ("decoder_fuse.d1_c1.conv.weight", "decoder_fuse.d1_c1.conv.bias", "decoder_fuse.d1_c2.conv.weight", "decoder_fuse.d2_c1.conv.bias", "decoder_fuse.d2_c1.conv.weight"):
    g = dict(model.named_parameters())[k].grad.cpu().double(); r = dict(ref.named_parameters())[k].grad
    print(k, "rel", rel(g, r), "norm", r.norm().item())
print("---- isolate the ReLU->IN backward kernel: recompute it in fp64 from HIP's own saved tensors")
for n in ("d1_c2", "d1_c1", "d2_c1"):
    xx = acts[n + ".conv"].detach().cpu().double()            # [B, D, H, W, C] conv output (pre-ReLU)
    dy = acts[n].grad.cpu().double()
    z = xx.clamp(min=0)
    mu = z.mean(dim=(1, 2, 3), keepdim=True); var = z.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt(); xh = (z - mu) * rstd
    dz = rstd * (dy - dy.mean(dim=(1, 2, 3), keepdim=True) - xh * (dy * xh).mean(dim=(1, 2, 3), keepdim=True))
    dx = dz * (xx > 0)
    got = acts[n + ".conv"].grad.cpu().double()
    print(n, "IN-bwd kernel rel err vs fp64 recomputation:", rel(got, dx), " strides of dy:", acts[n].grad.stride(), " rstd range", rstd.min().item(), rstd.max().item())
    print("    mean(dy)", dy.mean(dim=(1, 2, 3)).flatten()[:4].tolist(), " mean(dy*xh)", (dy * xh).mean(dim=(1, 2, 3)).flatten()[:4].tolist(), " |dy| max", dy.abs().max().item())
print("---- which input error matters for d1_c1's ReLU->IN backward")
def inbwd(xx, dy):
    z = xx.clamp(min=0)
    mu = z.mean(dim=(1, 2, 3), keepdim=True); var = z.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt(); xh = (z - mu) * rstd
    dz = rstd * (dy - dy.mean(dim=(1, 2, 3), keepdim=True) - xh * (dy * xh).mean(dim=(1, 2, 3), keepdim=True))
    return dz * (xx > 0)
for n in ("d1_c1", "d1_c2"):
    x_h = acts[n + ".conv"].detach().cpu().double(); dy_h = acts[n].grad.cpu().double()
    x_r = racts[n + ".conv"].detach().permute(0, 2, 3, 4, 1); dy_r = racts[n].grad.permute(0, 2, 3, 4, 1)
    x_3 = r32[n + ".conv"].detach().double().permute(0, 2, 3, 4, 1); dy_3 = r32[n].grad.double().permute(0, 2, 3, 4, 1)
    truth = inbwd(x_r, dy_r)
    print(n, "hip x + ref dy:", rel(inbwd(x_h, dy_r), truth), " ref x + hip dy:", rel(inbwd(x_r, dy_h), truth), " flips hip:", ((x_h > 0) != (x_r > 0)).float().mean().item(),
          "| cpu32 x + ref dy:", rel(inbwd(x_3, dy_r), truth), " ref x + cpu32 dy:", rel(inbwd(x_r, dy_3), truth), " flips cpu32:", ((x_3 > 0) != (x_r > 0)).float().mean().item())
    nz = dy_r.abs() > 1e-3 * dy_r.abs().max()
    print("    flips where |dy| is significant: hip", (((x_h > 0) != (x_r > 0)) & nz).sum().item(), "cpu32", (((x_3 > 0) != (x_r > 0)) & nz).sum().item(), "of", nz.sum().item())
print("---- raw conv outputs")
for n in ("d1_c1", "d1_c2", "d2_c1", "d1_out"):
    x_h = acts[n + ".conv"].detach().cpu().double(); x_r = racts[n + ".conv"].detach().permute(0, 2, 3, 4, 1); x_3 = r32[n + ".conv"].detach().double().permute(0, 2, 3, 4, 1)
    d = x_h - x_r
    print(n, "conv out rel: hip", rel(x_h, x_r), "cpu32", rel(x_3, x_r), " per-channel mean diff hip", d.mean(dim=(0, 1, 2, 3))[:4].tolist(), " rms", x_r.pow(2).mean().sqrt().item(),
          " rel of input: ", None)
    # is the error a scale? fit x_h = a * x_r per channel
    a = (x_h * x_r).sum(dim=(0, 1, 2, 3)) / (x_r * x_r).sum(dim=(0, 1, 2, 3))
    print("    per-channel scale fit a-1:", (a - 1)[:8].tolist())
print("---- error of d1_c1's conv output by depth plane (hip vs cpu32, relative to fp64)")
x_h = acts["d1_c1.conv"].detach().cpu().double(); x_r = racts["d1_c1.conv"].detach().permute(0, 2, 3, 4, 1); x_3 = r32["d1_c1.conv"].detach().double().permute(0, 2, 3, 4, 1)
for d in (0, 1, 2, 3, 8, 64, 126, 127):
    print("  d=%3d hip %.2e cpu32 %.2e   mean signed diff hip %.2e cpu32 %.2e" % (d, rel(x_h[:, d], x_r[:, d]), rel(x_3[:, d], x_r[:, d]), (x_h[:, d] - x_r[:, d]).mean().item(), (x_3[:, d] - x_r[:, d]).mean().item()))
dy_r = racts["d1_c1"].grad.permute(0, 2, 3, 4, 1)
print("  sum(dy_ref * (x_hip - x_ref)) =", (dy_r * (x_h - x_r)).sum().item(), "  sum(dy_ref * (x_cpu32 - x_ref)) =", (dy_r * (x_3 - x_r)).sum().item(), "  sum |dy*x| =", (dy_r * x_r).abs().sum().item())
print("---- per-channel statistics of relu(x) for d1_c1: hip / cpu32 relative to fp64")
def stats(xx, dy):
    z = xx.clamp(min=0)
    mu = z.mean(dim=(1, 2, 3)); var = z.var(dim=(1, 2, 3), unbiased=False); rstd = (var + 1e-5).rsqrt()
    xh = (z - mu[:, None, None, None]) * rstd[:, None, None, None]
    return mu.flatten(), rstd.flatten(), (dy * xh).mean(dim=(1, 2, 3)).flatten(), dy.mean(dim=(1, 2, 3)).flatten()
sr = stats(x_r, dy_r); sh = stats(x_h, dy_r); s3 = stats(x_3, dy_r)
for nm, i in (("mu", 0), ("rstd", 1), ("m2", 2)):
    print(" ", nm, "hip rel diff", ((sh[i] - sr[i]) / sr[i]).tolist()); print(" ", nm, "c32 rel diff", ((s3[i] - sr[i]) / sr[i]).tolist())
t = inbwd(x_r, dy_r); a = inbwd(x_h, dy_r); b = inbwd(x_3, dy_r)
e = (a - t)
print("  err energy by depth plane (hip):", [float(e[:, d].pow(2).sum() / e.pow(2).sum()) for d in (0, 1, 2, 3, 4, 8, 64)])
print("  err energy where masks differ (hip):", float((e * ((x_h > 0) != (x_r > 0))).pow(2).sum() / e.pow(2).sum()))
e3 = (b - t)
print("  err energy where masks differ (cpu32):", float((e3 * ((x_3 > 0) != (x_r > 0))).pow(2).sum() / e3.pow(2).sum()), " total err norms hip/cpu32:", e.norm().item(), e3.norm().item(), " truth norm", t.norm().item())
