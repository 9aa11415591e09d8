"""Time corrif_gemm_fwd / corrif_wgrad on a few shapes (GPU box).  Usage: python tools/gemm_microbench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, ops, corrif_hip as H
dev = "cuda:0"
def bench(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, K) in [(4096, 4096, 4096), (25088, 1024, 256), (25088, 256, 1024), (401408, 256, 64), (401408, 64, 256), (100352, 512, 128), (6272, 2048, 512), (16384, 1536, 512), (65536, 512, 512)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    ms = bench(lambda: ops.gemm(A.data_ptr(), K, B.data_ptr(), K, 0, C.data_ptr(), N, M, N, K, K, H.gemm_geom()))
    print("gemm  M %7d N %5d K %5d : %8.3f ms  %6.1f TF/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    Bt = torch.randn(K, N, device=dev)
    ms = bench(lambda: ops.gemm(A.data_ptr(), K, Bt.data_ptr(), N, 1, C.data_ptr(), N, M, N, K, K, H.gemm_geom()))
    print("bl1   M %7d N %5d K %5d : %8.3f ms  %6.1f TF/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
for (R, M, N) in [(25088, 256, 1024), (401408, 64, 256), (100352, 128, 512), (6272, 512, 2048), (401408, 256, 64), (100352, 512, 64)]:
    dY = torch.randn(R, M, device=dev); X = torch.randn(R, N, device=dev); dW = torch.empty(M, N, device=dev)
    ms = bench(lambda: ops.wgrad(dY.data_ptr(), M, X.data_ptr(), N, N, dW.data_ptr(), N, R, M, N, H.gemm_geom(), dev))
    print("wgrad R %7d M %5d N %5d : %8.3f ms  %6.1f TF/s" % (R, M, N, ms, 2.0 * R * M * N / ms / 1e9), flush=True)
