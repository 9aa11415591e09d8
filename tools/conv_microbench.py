"""Time corrif_gemm_fwd in implicit-conv mode against the same M/N/K as a plain GEMM (GPU box).
Usage: [CORRIF_LIB=...] python tools/conv_microbench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, ops, corrif_hip as H
dev = "cuda:0"
REPS = int(os.environ.get('REPS', '10'))
def bench(fn, n=REPS):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
# (B, D, H, W, Ci, Co, k)  encoder / decoder 3-D convolutions of the B=32 workload
for (B, D, Hh, W, Ci, Co, k) in [(32, 4, 14, 14, 256, 256, (1, 3, 3)), (32, 4, 28, 28, 128, 128, (1, 3, 3)), (32, 4, 7, 7, 512, 512, (1, 3, 3)),
                                 (32, 4, 56, 56, 64, 64, (1, 3, 3)), (32, 4, 28, 28, 128, 128, (3, 3, 3)), (32, 4, 56, 56, 32, 64, (3, 3, 3))][:int(os.environ.get('NSHAPES', '6'))]:
    M, K = B * D * Hh * W, k[0] * k[1] * k[2] * Ci
    x = torch.randn(M, Ci, device=dev); w = torch.randn(Co, K, device=dev); y = torch.empty(M, Co, device=dev)
    g = H.conv_geom((D, Hh, W), (D, Hh, W), k, (1, 1, 1), (k[0] // 2, k[1] // 2, k[2] // 2))
    ms = bench(lambda: ops.gemm(x.data_ptr(), Ci, w.data_ptr(), K, 0, y.data_ptr(), Co, M, Co, K, Ci, g))
    A = torch.randn(M, K, device=dev)
    ms2 = bench(lambda: ops.gemm(A.data_ptr(), K, w.data_ptr(), K, 0, y.data_ptr(), Co, M, Co, K, K, H.gemm_geom()))
    fl = 2.0 * M * Co * K / 1e9
    print("conv M %7d N %4d K %5d k%s: %7.3f ms %6.1f TF/s | as plain gemm %7.3f ms %6.1f TF/s" % (M, Co, K, k, ms, fl / ms, ms2, fl / ms2), flush=True)
    del A
