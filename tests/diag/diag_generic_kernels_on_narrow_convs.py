"""Diagnostic (GPU box): the narrow 3x3x3 layers through the implicit-GEMM kernels instead of the patch kernels (ops.USE_PATCH = False)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, torch.nn.functional as F, ops
import test_kernels_gpu as T
ops.USE_PATCH = False
DEV = "cuda:0"
for cfg in T.CONVS:
    name, Ci, Co, k, stride, pad, rep, (B, D, Hh, W), bias = cfg
    if k != (3, 3, 3) or "many" in name: continue
    x = T.rnd(B, Ci, D, Hh, W, seed=1); w = T.rnd(Co, Ci, *k, seed=2, scale=1.0 / math.sqrt(Ci * 27)); b = T.rnd(Co, seed=3) if bias else None
    xr, wr = x.clone().double().requires_grad_(), w.clone().double().requires_grad_()
    br = b.clone().double().requires_grad_() if bias else None
    yr = F.conv3d(F.pad(xr, (1,) * 6, mode="replicate"), wr, br, stride) if rep else F.conv3d(xr, wr, br, stride, pad)
    gy = T.rnd(*yr.shape, seed=4); yr.backward(gy.double())
    xg = T.cl(x).to(DEV).requires_grad_(); wg = w.to(DEV).requires_grad_(); bg = b.to(DEV).requires_grad_() if bias else None
    yg = ops.conv3d(xg, wg, bg, stride, pad, rep); yg.backward(T.cl(gy).to(DEV)); torch.cuda.synchronize()
    print("%-26s fwd %.2e dgrad %.2e wgrad %.2e" % (name, T.rel(T.ncdhw(yg).double(), yr), T.rel(T.ncdhw(xg.grad).double(), xr.grad), T.rel(wg.grad.double(), wr.grad)), flush=True)
