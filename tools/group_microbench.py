"""Does ONE grouped launch of the three modality encoders' identical-shape GEMMs (Z = 3), with and without the stream-K split, beat
three launches - back to back on one stream, and on three streams?  (GPU box; uses the Z batching corrif_gemm_fwd already has.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, ops, corrif_hip as H
dev = "cuda:0"
streams = [torch.cuda.Stream() for _ in range(3)]


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(name, M, N, K, conv=None):
    Ci = K if conv is None else conv[0]
    A = torch.randn(3, M, Ci, device=dev); W = torch.randn(3, N, K, device=dev); C = torch.empty(3, M, N, device=dev)
    g = H.gemm_geom() if conv is None else H.conv_geom(conv[1], conv[1], (1, 3, 3), (1, 1, 1), (0, 1, 1))
    fl = 3 * 2.0 * M * N * K

    def one(z):
        ops.gemm(A[z].data_ptr(), Ci, W[z].data_ptr(), K, 0, C[z].data_ptr(), N, M, N, K, Ci, g)

    def serial():
        for z in range(3):
            one(z)

    def three_streams():
        cur = torch.cuda.current_stream()
        for z in range(3):
            streams[z].wait_stream(cur)
            with torch.cuda.stream(streams[z]):
                one(z)
        for z in range(3):
            cur.wait_stream(streams[z])

    def grouped():
        ops.gemm(A.data_ptr(), Ci, W.data_ptr(), K, 0, C.data_ptr(), N, M, N, K, Ci, g, Z=3, sA=(M * Ci, 0), sB=(N * K, 0), sC=(M * N, 0))

    res = []
    for label, fn, sk, ks in (("3 launches, one stream", serial, False, True), ("same, ONE fma chain over K", serial, False, False),
                              ("3 launches, three streams", three_streams, False, True), ("one grouped launch (Z=3)", grouped, False, True),
                              ("grouped, ONE fma chain", grouped, False, False), ("grouped + stream-K", grouped, True, True)):
        ops.STREAM_K, ops.SPLIT_BF16 = sk, ks
        ms = timeit(fn)
        res.append("%s %.3f ms %.1f TF/s" % (label, ms, fl / ms / 1e9))
    ops.STREAM_K, ops.SPLIT_BF16 = False, True
    print("%-28s M=%6d N=%4d K=%4d | " % (name, M, N, K) + " | ".join(res), flush=True)


case("e4 conv2 (1x3x3 256)", 25088, 256, 2304, conv=(256, (4, 14, 14)))
case("e4 conv1 (1024->256)", 25088, 256, 1024)
case("e4 conv3 (256->1024)", 25088, 1024, 256)
case("e3 conv2 (1x3x3 128)", 100352, 128, 1152, conv=(128, (4, 28, 28)))
case("e3 conv1 (512->128)", 100352, 128, 512)
case("e3 conv3 (128->512)", 100352, 512, 128)
case("e5 conv2 (1x3x3 512)", 6272, 512, 4608, conv=(512, (4, 7, 7)))
case("e5 conv1 (2048->512)", 6272, 512, 2048)
case("e5 conv3 (512->2048)", 6272, 2048, 512)
case("e2 conv2 (1x3x3 64)", 401408, 64, 576, conv=(64, (4, 56, 56)))
case("e2 conv3 (64->256)", 401408, 256, 64)
case("e2 conv1 (256->64)", 401408, 64, 256)
