"""corrif_gemm_fwd on the shapes tools/gemm_lab.hip times (plain GEMM, with the BatchNorm-statistics epilogue, and as the 1x3x3 implicit
conv the encoder runs), to compare the product kernel with the lab's stripped main loop.  GPU box: python tools/gemm_vs_lab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, ops, corrif_hip as H
dev = "cuda:0"
def bench(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, K) in [(4096, 4096, 4096), (25088, 256, 2304), (6272, 512, 4608), (100352, 128, 1152), (25088, 1024, 256), (401408, 256, 64), (100352, 512, 128)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    ms = bench(lambda: ops.gemm(A.data_ptr(), K, B.data_ptr(), K, 0, C.data_ptr(), N, M, N, K, K, H.gemm_geom()))
    print("M %7d N %5d K %5d plain          : %8.3f ms  %6.1f TF/s" % (M, N, K, ms, fl / ms / 1e9), flush=True)
    chunks = (M + 63) // 64
    part = torch.empty(N * chunks * 2, dtype=torch.float64, device=dev)
    ms = bench(lambda: ops.gemm(A.data_ptr(), K, B.data_ptr(), K, 0, C.data_ptr(), N, M, N, K, K, H.gemm_geom(), stats=(part.data_ptr(), M, 0)))
    print("M %7d N %5d K %5d +stats         : %8.3f ms  %6.1f TF/s" % (M, N, K, ms, fl / ms / 1e9), flush=True)
# the same contraction sizes as the encoder's 1x3x3 convolutions (implicit gather)
for (B_, D, Hh, W, Ci, Co) in [(32, 4, 14, 14, 256, 256), (32, 4, 7, 7, 512, 512), (32, 4, 28, 28, 128, 128)]:
    x = torch.randn(B_, D, Hh, W, Ci, device=dev); w = torch.randn(Co, 9 * Ci, device=dev)
    y = torch.empty(B_, D, Hh, W, Co, device=dev)
    M = B_ * D * Hh * W
    geom = H.conv_geom((D, Hh, W), (D, Hh, W), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    ms = bench(lambda: ops.gemm(x.data_ptr(), Ci, w.data_ptr(), 9 * Ci, 0, y.data_ptr(), Co, M, Co, 9 * Ci, Ci, geom))
    print("conv 1x3x3 M %7d N %5d K %5d : %8.3f ms  %6.1f TF/s" % (M, Co, 9 * Ci, ms, 2.0 * M * Co * 9 * Ci / ms / 1e9), flush=True)
