"""Do two MFMA kernels co-running on two streams finish sooner than back-to-back on one stream?  (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, ops, corrif_hip as H
dev = "cuda:0"
def conv_case(B, D, Hh, W, Ci, Co, k):
    M, K = B * D * Hh * W, k[0] * k[1] * k[2] * Ci
    x = torch.randn(M, Ci, device=dev); w = torch.randn(Co, K, device=dev); y = torch.empty(M, Co, device=dev)
    g = H.conv_geom((D, Hh, W), (D, Hh, W), k, (1, 1, 1), (k[0] // 2, k[1] // 2, k[2] // 2))
    return (lambda: ops.gemm(x.data_ptr(), Ci, w.data_ptr(), K, 0, y.data_ptr(), Co, M, Co, K, Ci, g)), 2.0 * M * Co * K, (x, w, y)
def gemm_case(M, N, K):
    A = torch.randn(M, K, device=dev); Bm = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    return (lambda: ops.gemm(A.data_ptr(), K, Bm.data_ptr(), K, 0, C.data_ptr(), N, M, N, K, K, H.gemm_geom())), 2.0 * M * N * K, (A, Bm, C)
def norm_case(M, C):
    x = torch.randn(M, C, device=dev); y = torch.empty_like(x)
    return (lambda: torch.add(x, 1.0, out=y)), 0.0, (x, y)
cases = {"e4conv": lambda: conv_case(32, 4, 14, 14, 256, 256, (1, 3, 3)), "e3conv": lambda: conv_case(32, 4, 28, 28, 128, 128, (1, 3, 3)),
         "e2conv": lambda: conv_case(32, 4, 56, 56, 64, 64, (1, 3, 3)), "g1": lambda: gemm_case(25088, 1024, 256), "g2": lambda: gemm_case(100352, 512, 128),
         "mem": lambda: norm_case(401408 * 4, 64)}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(fa, fb, n, two):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record()
    if two:
        s1.wait_stream(cur); s2.wait_stream(cur)
        for _ in range(n):
            with torch.cuda.stream(s1): fa()
            with torch.cuda.stream(s2): fb()
        cur.wait_stream(s1); cur.wait_stream(s2)
    else:
        for _ in range(n): fa(); fb()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for a, b in [("e4conv", "e4conv"), ("e4conv", "e3conv"), ("e3conv", "e2conv"), ("e4conv", "g1"), ("g1", "g2"), ("e4conv", "mem"), ("e2conv", "mem")]:
    fa, fla, ka = cases[a](); fb, flb, kb = cases[b]()
    run(fa, fb, 3, False); run(fa, fb, 3, True)
    ser = run(fa, fb, 20, False); con = run(fa, fb, 20, True)
    print("%-7s + %-7s : serial %.3f ms (%.1f TF/s)   two streams %.3f ms (%.1f TF/s)" % (a, b, ser, (fla + flb) / ser / 1e9, con, (fla + flb) / con / 1e9), flush=True)
